#!/bin/bash
# Round 6: complete_pipeline with the from_wire leg (default workload and os1_128_cut3), driver form.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for w in stream100k os1_128_cut3; do
timeout 400 python bench.py --workload $w --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 > $O/bench_$w.json 2> $O/bench_$w.err
tail -3 $O/bench_$w.err
python -c "
import json; d=json.loads(open('$O/bench_$w.json').readline()); print('$w', round(d['value']), d['ms_per_step']); p=d['complete_pipeline']; print({k:v for k,v in p.items() if k!='from_wire' and k!='what'}); print(p.get('from_wire'))"
done
