#!/bin/bash
# map tests + the default bench with the complete pipeline.  usage: bash tools/gpu_pipe.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_register.py tests/test_gpu_scan_ops.py tests/test_gpu_multirank.py -q -x > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt | cut -c1-300
for rep in 1 2; do
LII_DIAG=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; grep "repeated\|rebuild" $O/bench.err | tail -2
python - <<PY
import json
d=json.loads(open('$O/bench.json').readline())
p=d['complete_pipeline']
print('value', round(d['value']), '| pipeline', round(p['value']), 'first pass', round(p['first_pass_growing_map']['value']), 'pageable', round(p['pageable_source']['value']), 'serial', round(p['serial_upload']['value']), 'map', p['map_points_after'])
PY
done
