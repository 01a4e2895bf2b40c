#!/bin/bash
# timelines of the default step and of the step with the map update.  usage: bash tools/gpu_tl.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for L in reg map; do
  X=""; [ $L = map ] && X="--map-update"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o $L -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline $X > $O/prof_$L.log 2>&1; echo "prof rc=$?"
  python tools/summarize_profile.py $O/prof_$L $O/${L}_kernel_summary.md "$L" > /dev/null 2>&1
  python tools/timeline.py $O/prof_$L $O/${L}_timeline.md "$L" > /dev/null 2>&1
  rm -rf $O/prof_$L
done
