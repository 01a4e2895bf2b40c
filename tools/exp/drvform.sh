cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('driver form', round(d['value']), d['ms_per_step'], 'long', round(d['value_long']['value']))"
done
