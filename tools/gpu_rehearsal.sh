#!/bin/bash
# One-device rehearsals of the sharded bench (ranks time-share device 0): usage gpurun -- 'bash tools/gpu_rehearsal.sh <outdir> [ranks...]'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
for n in "${@:-2 4}"; do
  LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 200 --no-cpu-baseline --no-calibration > $O/rehearsal_x$n.json 2> $O/rehearsal_x$n.err; echo "rehearsal x$n rc=$?"
  python - <<PY
import json
d=json.loads(open('$O/rehearsal_x$n.json').readlines()[-1])
print('x$n value', round(d['value']), 'transports', {k: round(v.get('value', 0)) for k, v in d['transports'].items()}, 'partitions', {k: round(v.get('value', 0)) for k, v in d['partitions'].items()})
PY
done
