#!/bin/bash
# Round 6: the solve's phase stamps (build_ab/solvetrace) and the device-side gap between two scans (build_ab/gap), unprofiled runs.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for v in solvetrace gap; do
  LII_LIB=$PWD/build_ab/$v/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/$v:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 400 --warmup 20 $COMMON > $O/$v.json 2> $O/$v.err
  echo "--- $v"; grep -a "solve trace\|gap trace" $O/$v.err | tail -6
  python -c "
import json; d=json.loads(open('$O/$v.json').readline()); print(round(d['value']), d['ms_per_step'])"
done
