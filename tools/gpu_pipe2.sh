#!/bin/bash
# map tests + the complete pipeline (default form and the driver's) after a change of the map update
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_end_to_end.py tests/test_replay_host.py tests/test_gpu_prearm.py tests/test_gpu_scan_ops.py -m gpu -q --timeout 300 2>&1 | grep -a -E "passed|failed" | tail -3
for f in "" "--steps 20 --warmup 5"; do
timeout 300 python bench.py $f --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 > $O/pipe.json 2> $O/pipe.err
python -c "
import json; d=json.loads(open('$O/pipe.json').readline()); p=d['complete_pipeline']; print('[$f]', round(d['value']), 'pipeline', round(p['value']), 'pageable', round(p['pageable_source']['value']), 'serial', round(p['serial_upload']['value']), 'wire', round(p['from_wire']['value']) if 'value' in p.get('from_wire',{}) else p.get('from_wire'))"
done
timeout 200 python bench.py --map-update --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 --steps 300 | cut -c1-120
