#!/bin/bash
# Round 3, first pass: the -m gpu suite on the new search kernel, then same-box A/B of the round-2 library against the new one
# (kernel variants, cell sizes).  usage: bash tools/gpu_r3a.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
L=lidar_imu_init_amd/lib
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() {  # name, env..., -- bench args
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").readline())
    print("$name", round(d["value"]), "scans/s  knn us", round(d["roofline"]["avg_launch_ms"]*1e3,2), "it", d["config"]["avg_iterations"], "knn passes", d["config"]["avg_knn_passes"])
except Exception as e:
    print("$name FAILED", e)
PY
}
cp $L/libliinit_hip.so $L/libliinit_hip.so.new
cp $L/libliinit_hip.so.r2 $L/libliinit_hip.so; EXTRA="" run r2 LII_X=0
cp $L/libliinit_hip.so.new $L/libliinit_hip.so
for v in 0 21 22 25 26 23 24 5; do EXTRA="" run v$v LII_KNN_VARIANT=$v; done
for cs in 0.5 0.55 0.6 0.7; do EXTRA="--cell-size $cs" run v0_cs$cs LII_KNN_VARIANT=0; EXTRA="--cell-size $cs" run v22_cs$cs LII_KNN_VARIANT=22; done
cp $L/libliinit_hip.so.r2 $L/libliinit_hip.so; EXTRA="" run r2_again LII_X=0
cp $L/libliinit_hip.so.new $L/libliinit_hip.so
EXTRA="" run v0_again LII_KNN_VARIANT=0
