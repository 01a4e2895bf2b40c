#!/bin/bash
# usage: bash tools/gpu_safe.sh <outdir> <per-test timeout s> <overall timeout s> <pytest args...>   (tight limits: a hung kernel costs GPU minutes)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; PT=$2; OT=$3; shift 3; mkdir -p $O
timeout $OT python -m pytest "$@" -m gpu -q -x --timeout $PT > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "thread" $O/pytest.log | tail -8
