#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_prearm.py tests/test_gpu_map.py -m gpu -q -x --timeout 300 2>&1 | tail -15
for v in 1 0 1 0; do
LII_PREARM=$v LII_DIAG=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 > $O/pipe_p$v.json 2> $O/pipe_p$v.err
python -c "
import json; d=json.loads(open('$O/pipe_p$v.json').readline()); p=d['complete_pipeline']; print('prearm=$v driver form', round(d['value']), 'pipeline', round(p['value']), 'pageable', round(p['pageable_source']['value']), 'serial', round(p['serial_upload']['value']), 'wire', round(p['from_wire']['value']) if 'value' in p.get('from_wire',{}) else p.get('from_wire'))"
grep -a "pre-armed" $O/pipe_p$v.err | tail -1
done
