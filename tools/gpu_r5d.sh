#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
LII_KNN_VARIANT=3 timeout 400 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_replay_host.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest(variant 3) rc=$?"; grep -E "C\+\+ host on|passed|failed|^E  " $O/pytest.log | cut -c1-400 | head -20
AB_PROFILE_ONLY=1 AB_TAG=pk_ bash tools/ab.sh r5d "tree" "stream100k dense500k vlp16"
AB_PROFILE_ONLY=1 AB_TAG=ck_ bash tools/ab.sh r5d "tree" "stream100k dense500k vlp16" LII_KNN_VARIANT=3
