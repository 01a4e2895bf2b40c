#!/bin/bash
# completion workgroups behind a search launch: 96 (tree) against 32 (build_ab/c32), plain stream and --edge, same box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_launch_plan.py tests/test_gpu_multirank.py -m gpu -q --timeout 300 2>&1 | grep -a -E "passed|failed|^FAILED" | tail -5
bash tools/ab.sh $1 "tree c32" "stream100k"
AB_ARGS=--edge AB_TAG=edge_ bash tools/ab.sh $1 "tree c32" "stream100k"
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
[ -f build_ab/fbtrace/libliinit_hip.so ] && LII_LIB=$PWD/build_ab/fbtrace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/fbtrace:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 200 --warmup 20 $COMMON 2>&1 >/dev/null | grep -a "completion trace"
