#!/bin/bash
# usage: bash tools/gpu_prof.sh <outdir> <label> [bench args...]  - kernel trace + summary + timeline of a bench run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; L=$2; shift; shift; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o $L -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 "$@" > $O/prof_$L.log 2>&1; echo "prof rc=$?"
python tools/summarize_profile.py $O/prof_$L $O/${L}_kernel_summary.md "$L" > /dev/null 2>&1
python tools/timeline.py $O/prof_$L $O/${L}_timeline.md "$L" > /dev/null 2>&1
rm -rf $O/prof_$L
cat $O/${L}_timeline.md | head -40
