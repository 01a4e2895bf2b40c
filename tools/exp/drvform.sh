cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
LII_STREAM_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 --profile-every 0 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('driver form', round(d['value']), d['ms_per_step'])"
grep -a "us per step" /tmp/e.txt | tail -3 | cut -c1-400
