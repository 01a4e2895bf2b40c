#!/bin/bash
# driver-form line three times on one box (+ debug stamps of the region's two ends)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0"
for i in 1 2 3; do
LII_BENCH_DEBUG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON > $O/drv$i.json 2> $O/drv$i.err
python -c "
import json; d=json.loads(open('$O/drv$i.json').readline()); print('driver form', round(d['value']), d['ms_per_step'], 'long', round(d['value_long']['value']), 'knn us', d['roofline'].get('avg_launch_ms'), d['roofline']['frac'])"
grep -a "bench debug\] timed" $O/drv$i.err | head -1
done
