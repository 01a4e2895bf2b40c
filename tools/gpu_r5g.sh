#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | cut -c1-300 | tail -20
bash tools/ab.sh r5g "tree" "stream100k"
AB_PROFILE_ONLY=1 bash tools/ab.sh r5g "tree" "dense500k vlp16 os1_128"
