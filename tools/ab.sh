#!/bin/bash
# A/B on the GPU box: for every variant (build_ab/<name>, or "tree" = the in-tree library) and workload, the kernel timeline of a
# short profiled run and the scans/s of an unprofiled one (AB_ARGS: more bench.py arguments, AB_TAG: a prefix of the labels).  usage: bash tools/ab.sh <outdir> "<variants>" "<workloads>" [env assignments for every run]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; VARS=$2; WLS=$3; shift; shift; shift; mkdir -p $O
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for v in $VARS; do
  if [ "$v" = "tree" ]; then unset LII_LIB; LDP=""; else export LII_LIB=$PWD/build_ab/$v/libliinit_hip.so; LDP=$PWD/build_ab/$v; fi
  for w in $WLS; do
    L=${AB_TAG}${v}_${w}
    if [ -z "$AB_PROFILE_ONLY" ]; then
      env "$@" LD_LIBRARY_PATH=$LDP:$LD_LIBRARY_PATH timeout 200 python bench.py --workload $w --steps 300 --warmup 30 $AB_ARGS $COMMON > $O/$L.json 2> $O/$L.err
    fi
    env "$@" LD_LIBRARY_PATH=$LDP:$LD_LIBRARY_PATH timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$L -o t -- python bench.py --workload $w --steps 80 --warmup 10 --prime 10 $AB_ARGS $COMMON > $O/prof_$L.log 2>&1
    python tools/timeline.py $O/prof_$L $O/${L}_timeline.md "$L" > /dev/null 2>&1
    rm -rf $O/prof_$L
    python tools/ab_line.py $O/$L.json $O/${L}_timeline.md "$L"
  done
done
