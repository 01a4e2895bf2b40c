#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for cs in 0.36 0.40 0.45 0.50 0.55; do
  timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pipeline --cell-size $cs > $O/bench_cs$cs.json 2> $O/bench_cs$cs.err
  python - $O/bench_cs$cs.json $cs <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("cs %s  scans/s %.0f  ms %.4f knn us %.1f"%(sys.argv[2],d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"]*1e3))
except Exception as e: print("  parse failed",e)
PY
done
