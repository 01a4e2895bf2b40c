#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python -m pytest tests/test_replay_host.py tests/test_gpu_register.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "per scan|C\+\+ host on|passed|failed" $O/pytest.log | cut -c1-700
bash tools/ab.sh r5c "tree" "stream100k"
bash tools/pmc_probe.sh r5c "stream100k dense500k"
grep -c . $O/counters_available.txt; grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*" $O/counters_available.txt | sort -u | tr '\n' ' ' | cut -c1-3000
