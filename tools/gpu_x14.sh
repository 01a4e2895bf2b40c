#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_x14.txt 2>&1; grep "passed\|failed" gpurun_out/pytest_x14.txt | tail -3
bash tools/gpu_prof.sh x14 early | grep "fit_reduce\|Scan period"
for rep in 1 2; do timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('default', round(d['value']))"; done
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); p=d['parity']; print('driver form', round(d['value']), p['dp_max'], p['dtheta_max'], p['dcov_gpu_vs_exact_max'])"
