#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
timeout 200 python tools/perscan.py > $O/perscan_tree.txt 2>&1; cat $O/perscan_tree.txt | tail -8
bash tools/ab.sh r5i "tree" "stream100k"
