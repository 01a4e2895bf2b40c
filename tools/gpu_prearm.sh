#!/bin/bash
# Round 6: the pre-armed prologue - its tests, then the driver-form line with and without the announcement (same box), the gap trace.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_prearm.py -m gpu -q -x --timeout 200 2>&1 | tail -15
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0"
for v in 1 0 1 0; do
LII_PREARM=$v LII_DIAG=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON > $O/drv_p$v.json 2> $O/drv_p$v.err
python -c "
import json; d=json.loads(open('$O/drv_p$v.json').readline()); print('prearm=$v driver form', round(d['value']), d['ms_per_step'], 'long', round(d['value_long']['value']), d['value_long']['ms_per_step'])"
grep -a "pre-armed" $O/drv_p$v.err | tail -1
done
