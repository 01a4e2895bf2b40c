#!/bin/bash
# usage: bash tools/pmc_diag.sh <outdir> <variant...> - SQ / TCP / TA counters of the k-NN kernel per variant (diagnostic)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --prime 0 --profile-every 0 --no-cpu-baseline --no-pipeline"
for v in "$@"; do
  i=0
  for SET in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
    i=$((i + 1))
    LII_KNN_VARIANT=$v timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/v$v/p$i" -o pmc -- $CMD > "$O/v$v.p$i.log" 2>&1 || echo "variant $v pass $i ($SET) failed"
  done
done
python - <<PY
import csv, glob, collections
for v in "$*".split():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/v%s/p*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_knn_pruned" not in r["Kernel_Name"]: continue
            per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    print("variant", v)
    for name in sorted(per):
        vals = sorted(per[name].values())
        big = [x for x in vals if x > 0.2 * vals[-1]] if vals and vals[-1] > 0 else vals
        print("  %-36s %14.0f  (%d launches)" % (name, sum(big) / max(len(big), 1), len(big)))
PY
