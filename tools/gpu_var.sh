#!/bin/bash
# usage: gpu_var.sh <outdir> <variant...> : parity tests + bench per LII_KNN_VARIANT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  LII_KNN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_map.py -m gpu -q -x --timeout 900 > $O/pytest_v$v.log 2>&1; echo "v$v pytest rc=$?"; tail -2 $O/pytest_v$v.log
  LII_KNN_VARIANT=$v timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pipeline > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - $O/bench_v$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  scans/s %.0f  ms %.4f knn us %.1f  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"]*1e3,d["roofline"]["frac"]))
except Exception as e: print("  parse failed",e)
PY
done
