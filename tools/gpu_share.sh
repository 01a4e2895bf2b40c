#!/bin/bash
# Kernel timeline of ONE rank's share of an N-rank job, in a single process (LII_TEST=solo_share): usage gpu_share.sh <outdir> <N...>  (N > 1 by voxel, N < -1 by index)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; shift
for n in "$@"; do
  LII_TEST=solo_share=$n bash tools/gpu_prof.sh $O share_$n --no-cpu-baseline > /dev/null 2>&1
  echo "== solo_share=$n"; sed -n 5,5p gpurun_out/$O/share_${n}_timeline.md; grep "^| [0-9]" gpurun_out/$O/share_${n}_timeline.md | cut -c1-70
done
