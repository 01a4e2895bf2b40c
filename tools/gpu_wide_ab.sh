#!/bin/bash
# Round 6: the listed-completion launch (k_complete_listed) on the --edge-wide / --edge / plain streams, on and off (LII_WIDE_COMPLETION), same box.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
X="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --steps 200"
for rep in 1 2; do
for w in "--edge-wide" "--edge" ""; do
for v in 1 0; do
  l=$(echo "x$w" | tr -d ' -')_$v
  LII_WIDE_COMPLETION=$v LII_DIAG=1 timeout 300 python bench.py $X $w ${PAR:-} > $O/$l.json 2> $O/$l.err
  python - <<PY
import json
d=json.loads(open('$O/$l.json').readline())
print('$w wide=$v', round(d['value']), d['ms_per_step'], 'long', round(d.get('value_long',{}).get('value',0)), d.get('edge',{}).get('unfinished_queries_per_search_pass'), d.get('parity',{}))
PY
  grep -a "unfinished queries:" $O/$l.err | tail -1
done; done; done
