#!/bin/bash
# copies the summaries of a tools/gpu_final.sh pass (gpurun_out/<dir>) into profiles/ under the round's names.  usage: bash tools/copy_final.sh <dir> <rNN>
S=gpurun_out/$1; R=$2; P=profiles
cp $S/bench_default.json $P/${R}_bench_default.json
cp $S/bench_driver_form.json $P/${R}_bench_driver_form.json
cp $S/${R}_bench_kernel_stats.csv $S/${R}_bench_kernel_summary.md $S/${R}_timeline.md $S/${R}_mapupdate_kernel_summary.md $S/${R}_mapupd_timeline.md $S/${R}_pmc_knn.json $P/
[ -f $S/perscan.txt ] && cp $S/perscan.txt $P/${R}_perscan.txt
for w in os1_128_cut3 stream100k; do [ -f $S/ingest_timeline_$w.md ] && cp $S/ingest_timeline_$w.md $P/${R}_ingest_timeline_$w.md; done
python3 - "$S" "$P/${R}_bench_edge.json" <<'PY'
import json, sys
s, out = sys.argv[1], sys.argv[2]
res = {}
for name, f in (("--edge", "bench_edge"), ("--edge-wide", "bench_edge-wide"), ("LII_PREARM=0", "bench_noprearm")):
    try:
        res[name] = json.loads(open(f"{s}/{f}.json").readline())
    except Exception as e:
        res[name] = {"error": str(e)}
json.dump(res, open(out, "w"), indent=1)
PY
grep -h -a "solve trace\|gap trace\|completion trace" $S/trace_*.err 2>/dev/null | sort | uniq -c | sort -rn | head -40 > $P/${R}_traces.txt
python3 - "$S" "$P/${R}_bench_other_workloads.json" <<'PY'
import json, sys
s, out = sys.argv[1], sys.argv[2]
res = {}
for name, f in (("vlp16", "bench_vlp16"), ("os1_128", "bench_os1_128"), ("os1_128_cut3", "bench_os1_128_cut3"), ("dense500k", "bench_dense500k"),
                ("--map-update", "bench_mapupdate"), ("LII_KNN_PLAN=0", "bench_noplan"), ("--upload", "bench_upload"), ("--no-downsample", "bench_nodown")):
    try:
        res[name] = json.loads(open(f"{s}/{f}.json").readline())
    except Exception as e:
        res[name] = {"error": str(e)}
json.dump(res, open(out, "w"), indent=1)
PY
for f in rehearsal_x2 rehearsal_x4 rehearsal_x8 rehearsal_cut3_x2; do [ -f $S/$f.json ] && tail -1 $S/$f.json > $P/${R}_rehearsal_one_device_${f#rehearsal_}.json; done
ls -la $P | grep $R
