#!/bin/bash
# Round 6: the completion path, short form - parity subset, the plain stream's timeline (window on), the phase stamps (window off / on).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py tests/test_gpu_launch_plan.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
AB_TAG=win1_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW=1
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for w in 0 1; do
  echo "--- completion trace, window $w"
  LII_WINDOW=$w LII_LIB=$PWD/build_ab/fbtrace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/fbtrace:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 200 --warmup 20 $COMMON 2>&1 >/dev/null | grep -a "completion trace"
done
