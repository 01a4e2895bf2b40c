#!/bin/bash
# Robustness pass: long soaks of the complete pipeline (plain, and with the late-publisher test hook: the placement-independent path of the
# in-launch prefix at length), repeated multi-rank rehearsals on one device.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak5; mkdir -p $O
timeout 300 python tools/soak_pipeline.py 150000 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -6 $O/soak.txt | cut -c1-300
LII_TEST=emit_late timeout 300 python tools/soak_pipeline.py 40000 > $O/soak_late.txt 2>&1; echo "soak (emit_late) rc=$?"; tail -3 $O/soak_late.txt | cut -c1-300
for r in 1 2 3; do
  LII_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2961$r bench.py --gpus 8 --steps 100 --no-cpu-baseline --no-calibration > $O/x8_$r.json 2> $O/x8_$r.err; echo "x8 run $r rc=$? $(tail -1 $O/x8_$r.json | cut -c1-90)"
done
