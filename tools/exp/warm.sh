cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for t in "" no_warm; do
LII_TEST=$t bash tools/gpu_prof.sh r4warm warm_$t --no-cpu-baseline > /dev/null 2>&1
echo "== [$t]"; sed -n 5,5p gpurun_out/r4warm/warm_${t}_timeline.md; grep "^| [0-9]" gpurun_out/r4warm/warm_${t}_timeline.md | cut -c1-60 | head -4; grep "^| 7 " gpurun_out/r4warm/warm_${t}_timeline.md | cut -c1-60
done
for t in "" no_warm "" no_warm; do
LII_TEST=$t timeout 300 python bench.py --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$t] default', round(d['value']), d['ms_per_step'], round(1e3*d['roofline']['avg_launch_ms'],2))"
done
