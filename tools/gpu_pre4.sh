#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_prearm.py tests/test_gpu_launch_plan.py -m gpu -q --timeout 300 2>&1 | grep -a -E "passed|failed" | tail -2
bash tools/gpu_drv3.sh $1
LII_STREAM_TRACE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 2>&1 >/dev/null | tail -c 220
