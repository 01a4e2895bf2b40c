#!/bin/bash
# Round 6: kernel timeline of complete_pipeline.from_wire (os1_128 layout, cut_frame_num 3) -> <outdir>/ingest_timeline.md
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for w in os1_128_cut3 stream100k; do
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_wire_$w -o t -- python bench.py --workload $w --steps 40 --warmup 5 --prime 10 --no-cpu-baseline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 > $O/prof_wire_$w.json 2> $O/prof_wire_$w.log
python tools/wire_timeline.py $O/prof_wire_$w $O/ingest_timeline_$w.md "complete_pipeline.from_wire, $w: the launches of one driver message" | head -70
rm -rf $O/prof_wire_$w
done
