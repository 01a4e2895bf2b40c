#!/bin/bash
# Round 5, first GPU pass: the suite on the in-tree library, then A/B of the round's builds.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
bash tools/ab.sh r5a "base solveonly tree" "stream100k"
bash tools/ab.sh r5a "base tree" "dense500k vlp16"
AB_PROFILE_ONLY=1 AB_TAG=lpq4_ bash tools/ab.sh r5a "tree" "stream100k vlp16 os1_128 os1_128_cut3 dense500k" LII_KNN_VARIANT=4
AB_PROFILE_ONLY=1 AB_TAG=lpq2_ bash tools/ab.sh r5a "tree" "stream100k vlp16 os1_128 os1_128_cut3 dense500k" LII_KNN_VARIANT=2
LII_LIB=$PWD/build_ab/trace/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/trace timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 > $O/trace.json 2> $O/trace.err; grep "solve trace" $O/trace.err | tail -3
