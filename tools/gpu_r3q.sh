#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log | cut -c1-200
for rep in 1 2; do for f in sort hash; do
  LII_VOXEL_FILTER=$f timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_$f.json 2> $O/bench_$f.err
  python -c "
import json; d=json.loads(open('$O/bench_$f.json').readline()); print('$f', round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
bash tools/gpu_prof.sh $1 hash 2>&1 | tail -20
