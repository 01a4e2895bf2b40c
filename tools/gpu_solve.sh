#!/bin/bash
# the solve after a change: its tests, the phase stamps (build_ab/solvetrace), the default step's timeline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_headline_parity.py tests/test_gpu_launch_plan.py tests/test_gpu_end_to_end.py tests/test_gpu_multirank.py -m gpu -q --timeout 300 2>&1 | tail -4
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
LII_LIB=$PWD/build_ab/solvetrace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/solvetrace:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 400 --warmup 20 $COMMON 2>&1 >/dev/null | grep -a "solve trace" | tail -3
bash tools/ab.sh $1 "tree" "stream100k"
