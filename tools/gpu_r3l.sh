#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for v in 0 42 50; do for dg in 256; do
  LII_KNN_VARIANT=$v LII_KNN_DIAG=$dg timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_${v}_$dg.json 2> $O/bench_${v}_$dg.err
  echo "variant $v diag $dg"; grep wlog $O/bench_${v}_$dg.err | cut -c1-200
  python -c "
import json; d=json.loads(open('$O/bench_${v}_$dg.json').readline()); print(round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
CMD="python bench.py --steps 8 --warmup 2 --prime 0 --profile-every 0 --no-cpu-baseline --no-pipeline"
for v in 0 42 50; do
  i=0
  for SET in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"; do
    i=$((i + 1))
    LII_KNN_VARIANT=$v timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/v$v/p$i" -o pmc -- $CMD > "$O/v$v.p$i.log" 2>&1 || echo "v $v pass $i ($SET) failed"
  done
done
python - <<PY
import csv, glob, collections
for v in "0 42 50".split():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/v%s/p*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_knn" not in r["Kernel_Name"]: continue
            per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    print("variant", v)
    for name in sorted(per):
        vals = sorted(per[name].values())
        big = [x for x in vals if x > 0.5 * vals[-1]] if vals and vals[-1] > 0 else vals
        print("  %-36s %14.0f  (%d launches)" % (name, sum(big) / max(len(big), 1), len(big)))
PY
rm -rf $O/v*/
