cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for f in "" "--map-update-separate"; do
LII_DIAG=1 timeout 300 python bench.py --steps 400 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 --map-update $f 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('map-update [$f]', round(d['value']), d['ms_per_step'], d['config']['avg_iterations'])"
grep -a "libliinit_hip" /tmp/e.txt | tail -4
done
