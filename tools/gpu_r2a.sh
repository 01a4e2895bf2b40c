#!/bin/bash
# first GPU pass of round 2: full -m gpu suite, then the k-NN variants A/B on the default bench step
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
for v in 64 4 65 32 128; do
  LII_KNN_VARIANT=$v LII_KNN_STATS=1 timeout 300 python bench.py --steps 200 --no-cpu-baseline > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "variant $v rc=$?"; python - $O/bench_v$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  scans/s %.0f  knn us %.1f  frac %.3f  n_d %.0f"%(d["value"],d["roofline"]["avg_launch_ms"]*1e3,d["roofline"]["frac"],d["config"]["downsampled_points"]))
except Exception as e: print("  parse failed",e)
PY
  grep -h "k_knn_tile workgroups" $O/bench_v$v.err | tail -1
done
LII_KNN_VARIANT=4 LII_VOXEL_ORDER=pcl timeout 300 python bench.py --steps 200 --no-cpu-baseline > $O/bench_v4_pcl.json 2> $O/bench_v4_pcl.err; echo "v4 pcl rc=$?"; cat $O/bench_v4_pcl.json | cut -c1-160
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; cat $O/bench_default.json
