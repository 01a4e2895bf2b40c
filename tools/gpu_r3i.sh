#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for v in 21; do for dg in 256; do
  LII_KNN_VARIANT=$v LII_KNN_DIAG=$dg timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${v}_$dg.json 2> $O/bench_${v}_$dg.err
  echo "variant $v diag $dg"; grep wlog $O/bench_${v}_$dg.err | cut -c1-200
done; done
