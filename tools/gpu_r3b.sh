#!/bin/bash
# ablation timings of the search kernel (results are wrong by construction: timing only)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").readline())
    print("$name", round(d["value"]), "scans/s  knn us", round(d["roofline"]["avg_launch_ms"]*1e3,2), "it", d["config"]["avg_iterations"], "knn passes", d["config"]["avg_knn_passes"])
except Exception as e:
    print("$name FAILED", e)
PY
}
for dg in 0 8 72 16 80 1 3 7 135 128 4 32 33 2; do EXTRA="" run diag$dg LII_KNN_DIAG=$dg; done
for dg in 0 8 16 135; do EXTRA="--workload vlp16" run vlp16_diag$dg LII_KNN_DIAG=$dg; done
