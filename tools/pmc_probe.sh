#!/bin/bash
# What bounds the search kernel: SQ issue / wait breakdown and the vector-memory front end (TA / TCP), one rocprofv3 pass per counter
# set (never with trace domains other than --kernel-trace).  usage: bash tools/pmc_probe.sh <outdir> "<workloads>" [env assignments]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; WLS=$2; shift; shift; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
CMDT="--steps 8 --warmup 2 --prime 0 --profile-every 0 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0"
for w in $WLS; do
  i=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
             "TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
             "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i + 1))
    env "$@" timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/${w}_p$i" -o pmc -- python bench.py --workload $w $CMDT > "$O/${w}_p$i.log" 2>&1 || echo "$w pass $i failed: $(tail -2 $O/${w}_p$i.log | head -1 | cut -c1-200)"
  done
  python tools/pmc_table.py $O $w
done
