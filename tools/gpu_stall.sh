#!/bin/bash
# Where does a one-off stall sit in the default bench region?  Per-step host times of the C++ loop (LII_STREAM_TRACE) of three fresh processes.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
F="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for r in 1 2 3; do
  LII_STREAM_TRACE=1 timeout 300 python bench.py $F > $O/run$r.json 2> $O/run$r.err
  python - $O/run$r.err $O/run$r.json <<'PY'
import sys, json, re
txt = open(sys.argv[1]).read()
rows = [l for l in txt.splitlines() if "[us per step" in l]
d = json.loads(open(sys.argv[2]).readline())
print("value", round(d["value"]), "slowest", d.get("slowest_step"))
base = 0
for l in rows:
    v = [float(x) for x in l.split("[")[0].split()]
    big = [(base + i, x) for i, x in enumerate(v) if x > 1000]
    print(" region of", len(v), "steps (from step", base, "): mean", round(sum(v) / len(v), 1), "steps > 1 ms:", big[:12])
    base += len(v)
PY
done
timeout 300 python bench.py > $O/default.json 2> $O/default.err; python -c "
import json; d=json.loads(open('$O/default.json').readline()); print('default', round(d['value']), d.get('slowest_step'), d['complete_pipeline']['value'])"
timeout 300 python tools/perscan.py 2>&1 | tail -8
