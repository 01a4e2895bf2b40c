#!/bin/bash
# whole GPU suite + driver-form line x2 + default line (round 6)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_full.txt 2>&1; grep -a -E "passed|failed|error|FAILED|ERROR|^E " $O/pytest_full.txt | tail -25 > $O/pytest.txt; cat $O/pytest.txt
bash tools/gpu_drv3.sh $1
