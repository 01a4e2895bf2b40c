#!/bin/bash
# usage: bash tools/gpu_pipe.sh <outdir>  - full GPU suite + default bench (with the complete-pipeline figures)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$O/bench.json").readline())
print(d["value"], d.get("complete_pipeline"))
PY
tail -3 $O/bench.err
