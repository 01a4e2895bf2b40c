#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --timeout 180 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pipeline --map-update > $O/bench_mapupd.json 2> $O/bench_mapupd.err; echo "map-update rc=$?"; cut -c1-130 $O/bench_mapupd.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_map -o t -- python bench.py --steps 60 --warmup 5 --prime 10 --no-cpu-baseline --no-pipeline --map-update > $O/prof_map.log 2>&1
python tools/summarize_profile.py $O/prof_map $O/summary_map.md "map-update" > /dev/null 2>&1
rm -rf $O/prof_map
cut -c1-120 $O/summary_map.md | sed -n 5,40p
