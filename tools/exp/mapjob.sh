cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_map.py tests/test_gpu_multirank.py -q -x 2>&1 | grep -a -E "passed|failed|Error|^E " | tail -15
for f in "" "--map-update-separate"; do
timeout 300 python bench.py --steps 400 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 --map-update $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('map-update [$f]', round(d['value']), d['ms_per_step'], d['config']['avg_iterations'])"
done
