#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
prof() { local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o t -- python bench.py --steps 60 --warmup 10 --prime 20 --no-cpu-baseline --profile-every 0 $EXTRA > $O/prof_$name.log 2>&1
  echo "prof $name rc=$?"
  python tools/timeline.py $O/prof_$name $O/timeline_$name.md "$name" > /dev/null 2>&1
  rm -rf $O/prof_$name
  sed -n 5,9p $O/timeline_$name.md; grep -E "k_solve_knn|k_reduce_solve|k_knn_pruned" $O/timeline_$name.md
}
EXTRA="--workload vlp16" prof vlp16_merged
EXTRA="--workload vlp16" prof vlp16_nomerge LII_NO_MERGE=1
EXTRA="--workload dense500k" prof dense_merged
