#!/bin/bash
# usage: bash tools/gpu_pmc_knn.sh <outdir> <diag...> - SQ counters of the search kernel (diagnostic)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --prime 0 --profile-every 0 --no-cpu-baseline --no-pipeline"
for v in "$@"; do
  i=0
  for SET in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_LEVEL_VMEM"; do
    i=$((i + 1))
    LII_KNN_DIAG=$v timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/v$v/p$i" -o pmc -- $CMD > "$O/v$v.p$i.log" 2>&1 || echo "diag $v pass $i ($SET) failed"
  done
done
python - <<PY
import csv, glob, collections
for v in "$*".split():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/v%s/p*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_knn" not in r["Kernel_Name"]: continue
            per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    print("diag", v)
    for name in sorted(per):
        vals = sorted(per[name].values())
        big = [x for x in vals if x > 0.2 * vals[-1]] if vals and vals[-1] > 0 else vals
        print("  %-36s %14.0f  (%d launches)" % (name, sum(big) / max(len(big), 1), len(big)))
PY
rm -rf $O/v*/
