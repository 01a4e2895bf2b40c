#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0"
LII_DIAG=1 LII_BENCH_DEBUG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $COMMON > $O/drv.json 2> $O/drv.err
grep -a "bench debug\|lii_synchronize" $O/drv.err | head -20
