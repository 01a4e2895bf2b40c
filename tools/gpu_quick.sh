#!/bin/bash
# usage: bash tools/gpu_quick.sh <outdir> <pytest args...>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
timeout 600 python -m pytest "$@" -m gpu -q --timeout 180 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
