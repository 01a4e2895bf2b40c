#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for v in 0 22 21; do for dg in 256 384; do
  LII_KNN_VARIANT=$v LII_KNN_DIAG=$dg timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${v}_$dg.json 2> $O/bench_${v}_$dg.err
  echo "variant $v diag $dg"; grep wlog $O/bench_${v}_$dg.err
  python -c "
import json; d=json.loads(open('$O/bench_${v}_$dg.json').readline()); print(round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
