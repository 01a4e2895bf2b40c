#!/bin/bash
# HBM traffic / L2 counters of the dominant kernel, one rocprofv3 pass per counter set (MI355X_MICROARCH.md: never mix --pmc
# with the sys/hip/hsa trace domains; --kernel-trace is fine).  Run ON THE GPU BOX from the repo root:
#   bash tools/collect_pmc.sh gpurun_out/pmc && python tools/summarize_pmc.py gpurun_out/pmc profiles/rNN_pmc_knn.json
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --steps 8 --warmup 2 --prime 0 --profile-every 0 --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
ls "$OUT"/p*/ | head -20
