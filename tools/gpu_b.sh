#!/bin/bash
# usage: bash tools/gpu_b.sh <outdir> [bench args]  - one default bench run, prints value + pipeline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
LII_KNN_STATS=1 timeout 600 python bench.py --steps 200 --warmup 20 "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$O/bench.json").readline())
print(d["value"], d.get("complete_pipeline"))
PY
grep libliinit $O/bench.err
