#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
AB_PROFILE_ONLY=1 bash tools/ab.sh r5f "tree nb4 nb8 bs256 bs64 wpe8 wpe6" "stream100k dense500k"
