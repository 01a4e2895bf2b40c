#!/bin/bash
# Round 6: per-scan kernel timeline with the map update in the job, with and without the two fusions (LII_MAP_FUSE).  usage: bash tools/gpu_mapprof.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for v in ${VARIANTS:-1 0}; do
  LII_MAP_FUSE=$v timeout 300 python bench.py --map-update --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --steps 200 > $O/bench_mapupdate_$v.json 2> $O/bench_mapupdate_$v.err; echo "fuse=$v map-update rc=$?"; cut -c1-130 $O/bench_mapupdate_$v.json
  LII_MAP_FUSE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_map_$v -o t -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 --map-update > $O/prof_map_$v.log 2>&1; echo "prof rc=$?"
  python tools/timeline.py $O/prof_map_$v $O/mapupd_timeline_$v.md "per-scan kernel timeline with the map update in the job (stream100k), LII_MAP_FUSE=$v" > /dev/null 2>&1
  sed -n 1,30p $O/mapupd_timeline_$v.md
  rm -rf $O/prof_map_$v
done
