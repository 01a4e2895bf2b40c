#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_prearm.py -m gpu -q --timeout 300 2>&1 | grep -a -E "passed|failed" | tail -2
bash tools/ab.sh $1 "tree c32" "stream100k"
