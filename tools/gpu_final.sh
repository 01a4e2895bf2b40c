#!/bin/bash
# Round-end measurement pass (run on the GPU box from the repo root): the -m gpu suite, the default bench line, the driver's form, kernel
# trace + stats + timeline of the same command, the PMC passes of the dominant kernel, the other workloads WITH the CPU baseline (every
# record carries `parity`), the map-update form, the one-device multi-rank rehearsals.  Everything lands under gpurun_out/$1; the
# summaries to keep are copied to profiles/ by hand.  usage: bash tools/gpu_final.sh <outdir> <rNN> [skip-list: words of tests pmc others map rehearsal]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; R=${2:-r06}; SKIP=" $3 "; mkdir -p $O
if [[ "$SKIP" != *" tests "* ]]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -6
fi
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cut -c1-400 $O/bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench driver-form rc=$?"; cut -c1-220 $O/bench_driver_form.json
CMDP="--steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $R -- python bench.py $CMDP > $O/prof.log 2>&1; echo "prof rc=$?"
python tools/summarize_profile.py $O/prof $O/${R}_bench_kernel_summary.md "Round 6 - python bench.py $CMDP (stream100k, voxel grid on, one lii_scan_register call per scan from the C++ host loop)" > /dev/null 2>&1
python tools/timeline.py $O/prof $O/${R}_timeline.md "Round 6 - per-scan kernel timeline of the default bench step (stream100k)" > /dev/null 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/${R}_bench_kernel_stats.csv 2>/dev/null
rm -rf $O/prof; tail -4 $O/${R}_timeline.md | cut -c1-600
if [[ "$SKIP" != *" pmc "* ]]; then
  bash tools/collect_pmc.sh $O/pmc > $O/pmc.log 2>&1; python tools/summarize_pmc.py $O/pmc $O/${R}_pmc_knn.json 25125988 > /dev/null 2>&1; rm -rf $O/pmc/p*/; python -c "
import json; d=json.load(open('$O/${R}_pmc_knn.json')); print({k: d[k] for k in ('hbm_bytes_per_launch','l2_hit_rate')}, d['counters_per_executed_launch'])"
fi
if [[ "$SKIP" != *" others "* ]]; then
  for w in vlp16 os1_128 os1_128_cut3 dense500k; do
    timeout 400 python bench.py --workload $w --no-pipeline --no-calibration --steps 200 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_$w.json').readline()); print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('parity',{}))" 2>&1 | cut -c1-700
  done
fi
if [[ "$SKIP" != *" others "* ]]; then
  X="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --steps 200"
  LII_KNN_PLAN=0 timeout 200 python bench.py $X > $O/bench_noplan.json 2> $O/bench_noplan.err; echo "LII_KNN_PLAN=0 rc=$?"; cut -c1-100 $O/bench_noplan.json
  timeout 200 python bench.py $X --upload > $O/bench_upload.json 2> $O/bench_upload.err; echo "--upload rc=$?"; cut -c1-100 $O/bench_upload.json
  timeout 200 python bench.py $X --no-downsample > $O/bench_nodown.json 2> $O/bench_nodown.err; echo "--no-downsample rc=$?"; cut -c1-100 $O/bench_nodown.json
fi
if [[ "$SKIP" != *" map "* ]]; then
  timeout 300 python bench.py --map-update --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --steps 200 > $O/bench_mapupdate.json 2> $O/bench_mapupdate.err; echo "map-update rc=$?"; cut -c1-130 $O/bench_mapupdate.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_map -o $R -- python bench.py $CMDP --map-update > $O/prof_map.log 2>&1; echo "prof map rc=$?"
  python tools/summarize_profile.py $O/prof_map $O/${R}_mapupdate_kernel_summary.md "Round 6 - python bench.py $CMDP --map-update (stream100k; the map update rides in the registration job)" > /dev/null 2>&1
  python tools/timeline.py $O/prof_map $O/${R}_mapupd_timeline.md "Round 6 - per-scan kernel timeline with the map update in the job (stream100k)" > /dev/null 2>&1
  rm -rf $O/prof_map
fi
if [[ "$SKIP" != *" rehearsal "* ]]; then
  for n in 2 4 8; do
    LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 200 --no-cpu-baseline --no-calibration > $O/rehearsal_x$n.json 2> $O/rehearsal_x$n.err; echo "rehearsal x$n rc=$?"; tail -1 $O/rehearsal_x$n.json | cut -c1-300
  done
  LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload os1_128_cut3 --steps 200 --no-cpu-baseline --no-calibration > $O/rehearsal_cut3_x2.json 2> $O/rehearsal_cut3_x2.err; echo "rehearsal cut3 x2 rc=$?"; tail -1 $O/rehearsal_cut3_x2.json | cut -c1-300
fi
timeout 200 python tools/perscan.py > $O/perscan.txt 2>&1; tail -8 $O/perscan.txt | cut -c1-60
if [[ "$SKIP" != *" edge "* ]]; then  # every scan looks past the map's edge (two sizes of the missing patch), CPU baseline + parity in the record
  for e in edge edge-wide; do
    timeout 400 python bench.py --$e --no-pipeline --no-calibration --no-live-traffic --steps 200 > $O/bench_$e.json 2> $O/bench_$e.err; echo "--$e rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_$e.json').readline()); print(round(d['value']), d['ms_per_step'], d['edge'])" 2>&1 | cut -c1-500
  done
  LII_PREARM=0 timeout 200 python bench.py --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --steps 400 > $O/bench_noprearm.json 2> $O/bench_noprearm.err; echo "LII_PREARM=0 rc=$?"; cut -c1-100 $O/bench_noprearm.json
fi
if [[ "$SKIP" != *" wire "* ]]; then  # complete_pipeline.from_wire: the launches of one driver message
  bash tools/gpu_wire2.sh $1 > $O/wire.log 2>&1; head -8 $O/ingest_timeline_os1_128_cut3.md | cut -c1-400
fi
if [[ "$SKIP" != *" traces "* ]]; then  # phase stamps (measurement builds under build_ab/: tools/ab_build.sh <name> - -DLII_..._TRACE)
  COMMON="--no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0"
  for v in solvetrace gap fbtrace; do
    [ -f build_ab/$v/libliinit_hip.so ] || continue
    for pre in 1 0; do
      [ $v != gap ] && [ $pre = 0 ] && continue
      LII_PREARM=$pre LII_LIB=$PWD/build_ab/$v/libliinit_hip.so LD_LIBRARY_PATH=$PWD/build_ab/$v:$LD_LIBRARY_PATH timeout 200 python bench.py --steps 400 --warmup 20 $COMMON > $O/trace_${v}_$pre.json 2> $O/trace_${v}_$pre.err
      echo "--- $v (LII_PREARM=$pre)"; grep -a "solve trace\|gap trace\|completion trace" $O/trace_${v}_$pre.err | tail -5 | cut -c1-700
    done
  done
fi
[[ "$SKIP" == *" second "* ]] && exit 0
timeout 600 python bench.py > $O/bench_default_2.json 2> $O/bench_default_2.err; echo "bench default (second run) rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default_2.json').readline()); print(round(d['value']), d.get('slowest_step'), round(d['complete_pipeline']['value']))"
