#!/bin/bash
# usage: bash tools/gpu_var.sh <outdir> <variant...>  - k-NN variants: parity tests + default bench per variant
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  LII_KNN_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_map.py tests/test_gpu_headline_parity.py -m gpu -q -x --timeout 600 > $O/pytest_v$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -1 $O/pytest_v$v.log
  LII_KNN_VARIANT=$v timeout 600 python bench.py --steps 200 --warmup 20 --no-pipeline --no-cpu-baseline > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_v$v.json").readline())
print("variant $v", d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done
