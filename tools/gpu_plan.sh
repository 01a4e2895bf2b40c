#!/bin/bash
# usage: bash tools/gpu_plan.sh <outdir>   - launch-plan test + headline parity + default bench with and without the plan
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_launch_plan.py tests/test_gpu_headline_parity.py tests/test_gpu_register.py tests/test_gpu_multirank.py -m gpu -q -s --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
LII_KNN_STATS=1 timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_plan.json 2> $O/bench_plan.err; echo "bench rc=$?"; head -c 400 $O/bench_plan.json; echo; grep parked $O/bench_plan.err
LII_KNN_PLAN=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-pipeline > $O/bench_noplan.json 2> $O/bench_noplan.err; echo "bench rc=$?"; head -c 200 $O/bench_noplan.json; echo
timeout 600 python bench.py --steps 200 --warmup 20 --map-update --no-pipeline > $O/bench_mapupd.json 2> $O/bench_mapupd.err; echo "bench rc=$?"; head -c 200 $O/bench_mapupd.json; echo
