#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_map.py tests/test_gpu_headline_parity.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
for v in 0 22 21 25 26 27 23 24; do for dg in 256; do
  LII_KNN_VARIANT=$v LII_KNN_DIAG=$dg timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${v}_$dg.json 2> $O/bench_${v}_$dg.err
  echo "variant $v diag $dg"; grep wlog $O/bench_${v}_$dg.err | cut -c1-200
  python -c "
import json; d=json.loads(open('$O/bench_${v}_$dg.json').readline()); print(round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
for v in 0 22; do
  LII_KNN_VARIANT=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${v}.json 2> $O/bench_${v}.err
  python -c "
import json; d=json.loads(open('$O/bench_${v}.json').readline()); print('variant $v no diag', round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done
