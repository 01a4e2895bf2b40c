#!/bin/bash
# Full GPU suite + default bench + driver-form bench.  Usage: gpurun -- 'bash tools/gpu_suite.sh <label>'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; grep -a -E "passed|failed|error|FAILED|ERROR|^E " $O/pytest_full.txt | tail -25 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').readline())
print('default', round(d['value']), d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline']['frac'], d.get('pipeline',{}))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_driver.json').readline()); print('driver form', round(d['value']), d['ms_per_step'])"
