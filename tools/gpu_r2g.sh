#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -h "device zero-phase" $O/pytest.log
timeout 300 python bench.py --steps 200 --no-cpu-baseline --map-update > $O/bench_mapupd.json 2> $O/bench_mapupd.err; echo "map-update rc=$?"; cut -c1-200 $O/bench_mapupd.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_map -o t -- python bench.py --steps 60 --warmup 5 --prime 10 --no-cpu-baseline --map-update > $O/prof_map.log 2>&1
python tools/summarize_profile.py $O/prof_map $O/summary_map.md "map-update" > /dev/null 2>&1; python tools/timeline.py $O/prof_map $O/timeline_map.md "map-update" > /dev/null 2>&1
rm -rf $O/prof_map
cat $O/summary_map.md | cut -c1-150
