#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline --map-update > $O/bench_map.json 2> $O/bench_map.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").readline()); print("default", round(d["value"]), d["complete_pipeline"]["value"], d["complete_pipeline"]["first_pass_growing_map"]["value"])
d=json.loads(open("$O/bench_map.json").readline()); print("map-update", round(d["value"]))
PY
