#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_map.py tests/test_gpu_headline_parity.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log | cut -c1-200
run() {
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
for v in 0 22 21 42; do EXTRA="" run v$v LII_KNN_VARIANT=$v; done
for cs in 0.36 0.4 0.5 0.55; do for v in 0 22; do EXTRA="--cell-size $cs" run v${v}_cs$cs LII_KNN_VARIANT=$v; done; done
