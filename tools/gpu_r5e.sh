#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|LO on whole|C\+\+ host on" $O/pytest.log | cut -c1-300 | tail -20
bash tools/ab.sh r5e "tree" "stream100k"
AB_PROFILE_ONLY=1 bash tools/ab.sh r5e "tree" "dense500k vlp16"
for i in 1 2; do
  LII_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2960$i bench.py --gpus 8 --steps 100 --no-cpu-baseline --no-calibration > $O/x8_$i.json 2> $O/x8_$i.err; echo "x8 run $i rc=$?"; tail -1 $O/x8_$i.json | cut -c1-260
done
