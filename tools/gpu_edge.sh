#!/bin/bash
# Round 6: the completion path (queries that look past the map's edge) - parity of the tree, then window on / off on the plain stream and
# on the --edge stream, then the phase stamps of the completion (build_ab/fbtrace = the tree + -DLII_FALLBACK_TRACE).
# usage: bash tools/gpu_edge.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py tests/test_gpu_launch_plan.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
AB_TAG=win0_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW=0
AB_TAG=win1_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW=1
AB_ARGS=--edge AB_TAG=edge_win0_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW=0
AB_ARGS=--edge AB_TAG=edge_win1_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW=1
COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
for w in 0 1; do for e in "" "--edge"; do
  echo "--- completion trace, window $w $e"
  LII_WINDOW=$w LII_LIB=$PWD/build_ab/fbtrace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/fbtrace:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 200 --warmup 20 $e $COMMON 2>&1 >/dev/null | grep -a "completion trace"
done; done
timeout 300 python bench.py --edge --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic > $O/bench_edge.json 2> $O/bench_edge.err
python -c "
import json; d=json.loads(open('$O/bench_edge.json').readline()); print('edge line', round(d['value']), d['ms_per_step'], d['edge'])"
