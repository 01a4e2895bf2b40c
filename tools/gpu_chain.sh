#!/bin/bash
# Round 6: prices of the levers on the scan's dependent chain (profiles/r06_chain.md).  Parity tests of the tree first (the dense cell
# window is on by default), then same-box A/B lines + timelines: window off / on, the gap-trace build, the emit without its in-launch prefix.
# usage: bash tools/gpu_chain.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_register.py tests/test_gpu_full_size.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py tests/test_gpu_launch_plan.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5
AB_TAG=win0_ bash tools/ab.sh $1 "tree" "stream100k dense500k vlp16" LII_WINDOW=0
AB_TAG=win1_ bash tools/ab.sh $1 "tree" "stream100k dense500k vlp16" LII_WINDOW=1
bash tools/ab.sh $1 "gap noprefix" "stream100k"
grep -h "gap trace" $O/*.err | tail -3
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; cut -c1-200 $O/bench_driver_form.json
