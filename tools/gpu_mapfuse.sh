#!/bin/bash
# Round 6: the map update's two fusions (hash insert in k_map_decide, the inserts' cells in the fold launch) - the map tests of the tree, then
# complete_pipeline with and without (LII_MAP_FUSE=0), same box.  usage: bash tools/gpu_mapfuse.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_window.py tests/test_gpu_first_divergence.py tests/test_gpu_prearm.py tests/test_gpu_end_to_end.py tests/test_gpu_scan_ops.py -m gpu -q -x --timeout 300 > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt | cut -c1-300
LII_TEST=pred_small timeout 600 python -m pytest tests/test_gpu_map.py tests/test_gpu_window.py -m gpu -q -x --timeout 300 > $O/pytest_small.txt 2>&1; tail -3 $O/pytest_small.txt | cut -c1-300
for v in 1 0 1 0; do
LII_MAP_FUSE=$v LII_DIAG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err; grep "repeated\|rebuild" $O/bench_$v.err | tail -2
python - <<PY
import json
d=json.loads(open('$O/bench_$v.json').readline())
p=d['complete_pipeline']; w=p.get('from_wire',{})
print('fuse=$v value', round(d['value']), '| pipeline', round(p['value']), 'first pass', round(p['first_pass_growing_map']['value']), 'pageable', round(p['pageable_source']['value']), 'serial', round(p['serial_upload']['value']), 'wire', round(w.get('value',0)), 'map', p['map_points_after'])
PY
done
