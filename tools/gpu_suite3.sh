#!/bin/bash
# whole GPU suite, per-scan host times, driver form x3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_full.txt 2>&1; grep -a -E "passed|failed|error|FAILED|ERROR|^E " $O/pytest_full.txt | tail -12
timeout 200 python tools/perscan.py 2>&1 | tail -8 | cut -c1-80
bash tools/gpu_drv3.sh $1
