#!/bin/bash
# Round 6: the overlapped ingest (ABI 9) - from_wire under variants of the ingest streams (priority, one / two streams, queue count)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
run() {  # label, env...
  local l=$1; shift
  env "$@" timeout 300 python bench.py --workload ${W:-stream100k} --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 > $O/wire_$l.json 2> $O/wire_$l.err
  python - <<PY
import json
d=json.loads(open('$O/wire_$l.json').readline()); p=d['complete_pipeline']; w=p.get('from_wire',{})
g=lambda r: (round(r['value']), round(r['ingest_us_per_message'])) if isinstance(r,dict) and 'value' in r else r
print('$l', 'value', round(d['value']), 'pipeline', round(p['value']), 'wire: overlapped', g(w), 'pageable', g(w.get('pageable_source')), 'begun early', g(w.get('begun_before_the_registrations')), 'behind', g(w.get('begun_behind_the_registrations')), 'serial', g(w.get('serial_ingest')))
PY
}
run default A=1
run default2 A=1
W=os1_128_cut3 run cut3 A=1
