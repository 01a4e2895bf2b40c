#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline "$@" > $O/prof.log 2>&1; echo "prof rc=$?"
python tools/summarize_profile.py $O/prof $O/summary.md "x" > /dev/null 2>&1
python tools/timeline.py $O/prof $O/timeline.md "x" > /dev/null 2>&1
rm -rf $O/prof
head -12 $O/summary.md | cut -c1-150; cat $O/timeline.md | cut -c1-100
