#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for w in 1; do
  echo "--- completion trace, window $w"
  LII_WINDOW=$w LII_LIB=$PWD/build_ab/fbtrace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/fbtrace:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 200 --warmup 20 $COMMON 2>&1 >/dev/null | grep -a "completion trace"
done
