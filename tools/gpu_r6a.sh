#!/bin/bash
# Round 6, first look: the full GPU suite of the tree, then where the driver-form region (20 steps) loses time against the long one.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; grep -a -E "passed|failed|error|FAILED|ERROR|^E " $O/pytest_full.txt | tail -25 > $O/pytest.txt; cat $O/pytest.txt
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0"
for i in 1 2 3; do
LII_BENCH_DEBUG=1 LII_DIAG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON > $O/drv$i.json 2> $O/drv$i.err
python -c "
import json; d=json.loads(open('$O/drv$i.json').readline()); print('driver form', round(d['value']), d['ms_per_step'], d.get('value_long'), d.get('slowest_step'))"
grep -a "bench debug\|libliinit_hip\]" $O/drv$i.err | tail -8
done
LII_STREAM_TRACE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON --long-steps 0 > $O/drv_trace.json 2> $O/drv_trace.err
tail -c 1500 $O/drv_trace.err
