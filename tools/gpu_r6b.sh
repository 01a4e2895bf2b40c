#!/bin/bash
# Round 6: driver-form line with the k-NN events inside the dispatch and the polled synchronise, against the bracket form (same box).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0"
for v in 0 1 0 1; do
LII_PROF_BRACKET=$v LII_BENCH_DEBUG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON > $O/drv_b$v.json 2> $O/drv_b$v.err
python -c "
import json; d=json.loads(open('$O/drv_b$v.json').readline()); print('bracket=$v driver form', round(d['value']), d['ms_per_step'], 'long', round(d['value_long']['value']), 'knn ev us', d['roofline'].get('avg_launch_ms'), d['roofline']['frac'])"
grep -a "bench debug" $O/drv_b$v.err | head -2
done
LII_STREAM_TRACE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON --long-steps 0 > $O/drv_trace.json 2> $O/drv_trace.err
tail -c 300 $O/drv_trace.err
timeout 600 python -m pytest tests/test_gpu_register.py tests/test_gpu_launch_plan.py -m gpu -q -x 2>&1 | tail -3
