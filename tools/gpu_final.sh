#!/bin/bash
# Round-end measurement pass (run on the GPU box from the repo root): the -m gpu suite, the default bench line, kernel trace
# + stats + timeline of the same command, the PMC passes of the dominant kernel, the other workloads / options, the
# one-device multi-rank rehearsals.  Everything lands under gpurun_out/$1; the summaries to keep are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; R=${2:-r04}; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --timeout 180 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cat $O/bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench driver-form rc=$?"; cut -c1-220 $O/bench_driver_form.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $R -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 > $O/prof.log 2>&1; echo "prof rc=$?"
python tools/summarize_profile.py $O/prof $O/${R}_bench_kernel_summary.md "Round 4 - python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 (stream100k, voxel grid on, one lii_scan_register call per scan from the C++ host loop)" > /dev/null 2>&1
python tools/timeline.py $O/prof $O/${R}_timeline.md "Round 4 - per-scan kernel timeline of the default bench step (stream100k)" > /dev/null 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/${R}_bench_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
bash tools/collect_pmc.sh $O/pmc > $O/pmc.log 2>&1; python tools/summarize_pmc.py $O/pmc $O/${R}_pmc_knn.json 25125988 > /dev/null 2>&1; rm -rf $O/pmc/p*/; cat $O/${R}_pmc_knn.json | head -40
for w in vlp16 os1_128 os1_128_cut3 dense500k; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-pipeline --no-calibration --steps 200 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; cut -c1-130 $O/bench_$w.json
done
timeout 300 python bench.py --map-update --no-cpu-baseline --no-pipeline --no-calibration --steps 200 > $O/bench_mapupdate.json 2> $O/bench_mapupdate.err; echo "map-update rc=$?"; cut -c1-130 $O/bench_mapupdate.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_map -o $R -- python bench.py --steps 100 --warmup 10 --prime 20 --map-update --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 > $O/prof_map.log 2>&1; echo "prof map rc=$?"
python tools/summarize_profile.py $O/prof_map $O/${R}_mapupdate_kernel_summary.md "Round 4 - python bench.py --steps 100 --warmup 10 --prime 20 --map-update --no-cpu-baseline --no-pipeline (stream100k; lii_scan_register + lii_map_incremental per scan: in-place map update)" > /dev/null 2>&1
rm -rf $O/prof_map
LII_KNN_PLAN=0 timeout 300 python bench.py --no-cpu-baseline --no-pipeline --no-calibration --steps 200 > $O/bench_noplan.json 2> $O/bench_noplan.err; echo "no launch plan rc=$?"; cut -c1-130 $O/bench_noplan.json
timeout 300 python bench.py --upload --no-cpu-baseline --no-pipeline --no-calibration --steps 200 > $O/bench_upload.json 2> $O/bench_upload.err; echo "upload rc=$?"; cut -c1-130 $O/bench_upload.json
timeout 300 python bench.py --no-downsample --no-cpu-baseline --no-pipeline --no-calibration --steps 200 > $O/bench_nodown.json 2> $O/bench_nodown.err; echo "no-downsample rc=$?"; cut -c1-130 $O/bench_nodown.json
for n in 2 4; do
  LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 200 --no-cpu-baseline --no-calibration > $O/rehearsal_x$n.json 2> $O/rehearsal_x$n.err; echo "rehearsal x$n rc=$?"; tail -1 $O/rehearsal_x$n.json | cut -c1-400
done
LII_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload os1_128_cut3 --steps 200 --no-cpu-baseline --no-calibration > $O/rehearsal_cut3_x2.json 2> $O/rehearsal_cut3_x2.err; echo "rehearsal cut3 x2 rc=$?"; tail -1 $O/rehearsal_cut3_x2.json | cut -c1-300
