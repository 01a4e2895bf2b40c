#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for dg in 256 384 264 272 391; do
  LII_KNN_DIAG=$dg timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_$dg.json 2> $O/bench_$dg.err
  echo "diag $dg"; grep wlog $O/bench_$dg.err
  python -c "
import json; d=json.loads(open('$O/bench_$dg.json').readline()); print(round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done
LII_KNN_DIAG=256 timeout 300 python bench.py --workload vlp16 --steps 100 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_vlp.json 2> $O/bench_vlp.err; echo vlp16; grep wlog $O/bench_vlp.err
LII_KNN_DIAG=256 timeout 300 python bench.py --workload dense500k --steps 50 --warmup 10 --no-cpu-baseline --no-pipeline > $O/bench_d5.json 2> $O/bench_d5.err; echo dense500k; grep wlog $O/bench_d5.err
