#!/bin/bash
# the dense window kept through map updates: tests, then the pipeline with LII_WINDOW_KEEP=1 / 0 (same box)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_map.py tests/test_gpu_end_to_end.py tests/test_replay_host.py tests/test_gpu_first_divergence.py tests/test_gpu_multirank.py -m gpu -q --timeout 400 2>&1 | grep -a -E "passed|failed|^FAILED|^E  " | tail -8
for v in 1 0 1 0; do
LII_WINDOW_KEEP=$v LII_DIAG=1 timeout 300 python bench.py --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 > $O/pipe_w$v.json 2> $O/pipe_w$v.err
python -c "
import json; d=json.loads(open('$O/pipe_w$v.json').readline()); p=d['complete_pipeline']; print('keep=$v', round(d['value']), 'pipeline', round(p['value']), 'first pass', round(p['first_pass_growing_map']['value']), 'wire', round(p['from_wire']['value']) if 'value' in p.get('from_wire',{}) else p.get('from_wire'))"
grep -a "dense cell window" $O/pipe_w$v.err | tail -1
done
for v in 1 0; do LII_WINDOW_KEEP=$v timeout 200 python bench.py --map-update --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 --steps 300 | cut -c1-110; done
