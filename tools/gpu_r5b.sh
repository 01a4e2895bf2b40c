#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
./tools/ubench/solve_ubench > $O/solve_ubench.txt 2>&1; cat $O/solve_ubench.txt
timeout 600 python -m pytest tests/test_gpu_register.py tests/test_gpu_headline_parity.py tests/test_replay_host.py tests/test_gpu_end_to_end.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
bash tools/ab.sh r5b "base tree" "stream100k dense500k os1_128"
AB_PROFILE_ONLY=1 AB_TAG=lpq4_ bash tools/ab.sh r5b "tree" "dense500k" LII_KNN_VARIANT=4
