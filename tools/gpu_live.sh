#!/bin/bash
# The driver's form of the bench with the live PMC passes, timed end to end.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/live; mkdir -p $O
T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err; echo "rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s"
python -c "
import json; d=json.loads(open('$O/driver_form.json').readline()); r=d['roofline']; print(round(d['value']), r['traffic'], r['alg_bytes_per_launch'], r['traffic_source'])"
T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $O/driver_form_nolive.json 2> $O/driver_form_nolive.err; echo "rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s (without the live passes)"
