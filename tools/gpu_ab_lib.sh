#!/bin/bash
# A/B of several builds of the library on the same box: usage  bash tools/gpu_ab_lib.sh <outdir> <suffix...>
# compares lidar_imu_init_amd/lib/libliinit_hip.so.<suffix> (copied over libliinit_hip.so in turn; the original is restored)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
L=lidar_imu_init_amd/lib
cp $L/libliinit_hip.so $L/libliinit_hip.so.orig
for rep in 1 2; do
for which in "$@"; do
  cp $L/libliinit_hip.so.$which $L/libliinit_hip.so
  timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${which}.json 2> $O/bench_${which}.err; python - <<PY
import json
d=json.loads(open("$O/bench_${which}.json").readline())
print("$which", round(d["value"]), round(d["roofline"]["avg_launch_ms"]*1e3,2))
PY
done
done
cp $L/libliinit_hip.so.orig $L/libliinit_hip.so
