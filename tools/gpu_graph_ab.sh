#!/bin/bash
# plain launches against the hipGraph replay of the update passes (LII_TEST=graph), same box.  usage: bash tools/gpu_graph_ab.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
LII_TEST=graph timeout 600 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_launch_plan.py tests/test_gpu_scan_ops.py -q -x 2>&1 | tail -3
for rep in 1 2; do for t in none graph; do
  LII_TEST=$t timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-pipeline > $O/b_$t.json 2>/dev/null
  LII_TEST=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $O/d_$t.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/b_$t.json').readline()); e=json.loads(open('$O/d_$t.json').readline()); print('$t: 300 steps', round(d['value']), 'knn us', round(d['roofline']['avg_launch_ms']*1e3,2), '| driver form', round(e['value']))"
done; done
for t in none graph; do
  LII_TEST=$t LII_DIAG=1 timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-pipeline 2>&1 >/dev/null | grep "host side"
  LII_TEST=$t timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -o $t -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline > $O/prof_$t.log 2>&1
  python tools/timeline.py $O/prof_$t $O/${t}_timeline.md "$t" > /dev/null 2>&1; rm -rf $O/prof_$t
  grep "Scan period" $O/${t}_timeline.md
done
