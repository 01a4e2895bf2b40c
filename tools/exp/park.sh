cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_launch_plan.py tests/test_gpu_stream_driver.py tests/test_gpu_end_to_end.py -q -x 2>&1 | grep -a -E "passed|failed|Error|^E " | tail -8
for i in 1 2; do
LII_DIAG=1 timeout 300 python bench.py --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('default', round(d['value']), d['ms_per_step'], d['config']['avg_iterations'], d['config']['avg_knn_passes'])"
grep -a "parked" /tmp/e.txt | tail -1
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('driver form', round(d['value']), 'long', round(d['value_long']['value']))"
