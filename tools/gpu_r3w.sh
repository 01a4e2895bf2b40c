#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for w in os1_128 dense500k os1_128_cut3 vlp16; do for v in 0 22 21; do
  LII_KNN_VARIANT=$v timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline > $O/b_${w}_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/b_${w}_$v.json').readline()); print('$w variant $v', round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2), d['config']['downsampled_points'])"
done; done
