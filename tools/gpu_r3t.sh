#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x --timeout 300 > $O/pytest_mr.log 2>&1; echo "pytest multirank rc=$?"; grep "passed\|failed" $O/pytest_mr.log | cut -c1-200; grep -n "^E " $O/pytest_mr.log | head -10
for n in 2; do
  LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 200 --no-cpu-baseline > $O/rehearsal_x$n.json 2> $O/rehearsal_x$n.err; echo "rehearsal x$n rc=$?"; tail -1 $O/rehearsal_x$n.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value']), d['config']['parallelism']); print(json.dumps(d.get('transports'),indent=0)[:900])"
  tail -3 $O/rehearsal_x$n.err
done
