#!/bin/bash
# Same-box A/B of the working tree against build_ab/prev: parity tests of the tree, per-scan host times (the edge scan), timelines.
# usage: bash tools/gpu_ab5.sh <outdir> "<workloads>" [pytest files]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
T=${3:-tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py tests/test_gpu_launch_plan.py}
timeout 500 python -m pytest $T -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
LII_LIB=$PWD/build_ab/prev/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/prev:$LD_LIBRARY_PATH timeout 200 python tools/perscan.py 2>&1 | tail -8 | cut -c1-20 | tr '\n' ' '; echo " (prev)"
timeout 200 python tools/perscan.py 2>&1 | tail -8 | cut -c1-20 | tr '\n' ' '; echo " (tree)"
bash tools/ab.sh $1 "prev tree" "$2"
