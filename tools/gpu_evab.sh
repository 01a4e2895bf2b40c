#!/bin/bash
# A/B of the profiling events' flags: default bench regions (every 8th step bracketed) with and without the system-scope fence of the events.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evab; mkdir -p $O
F="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for v in evfence tree evfence tree; do
  if [ "$v" = "tree" ]; then unset LII_LIB; LDP=""; else export LII_LIB=$PWD/build_ab/$v/libliinit_hip.so; LDP=$PWD/build_ab/$v; fi
  for pe in 8 0; do
    LD_LIBRARY_PATH=$LDP:$LD_LIBRARY_PATH timeout 200 python bench.py $F --profile-every $pe > $O/$v.$pe.json 2> $O/$v.$pe.err
    python -c "
import json; d=json.loads(open('$O/$v.$pe.json').readline()); print('$v profile-every $pe:', round(d['value']), 'scans/s, slowest', d['slowest_step']['ms'], d['slowest_step']['second_ms'], 'k-NN by events', d['roofline']['avg_launch_ms'])"
  done
done
