#!/bin/bash
# A/B of two builds of the library on the same box: lib/libliinit_hip.so (new) against lib/libliinit_hip.so.old
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
L=lidar_imu_init_amd/lib
cp $L/libliinit_hip.so $L/libliinit_hip.so.new
for rep in 1 2; do
for which in new old; do
  cp $L/libliinit_hip.so.$which $L/libliinit_hip.so
  for w in stream100k vlp16; do
    timeout 600 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline "$@" > $O/bench_${which}_$w.json 2> $O/bench_${which}_$w.err; python - <<PY
import json
d=json.loads(open("$O/bench_${which}_$w.json").readline())
print("$which $w", round(d["value"]), round(d["roofline"]["avg_launch_ms"]*1e3,2))
PY
  done
done
done
cp $L/libliinit_hip.so.new $L/libliinit_hip.so
