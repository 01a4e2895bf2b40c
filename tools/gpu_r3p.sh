#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log | cut -c1-200
LII_FUSE_TAIL=0 bash tools/gpu_prof.sh $1 unfused 2>&1 | tail -20
