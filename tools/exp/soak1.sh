cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
LII_DIAG=1 timeout 600 python bench.py --steps 20000 --map-update --long-steps 0 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('soak map-update', round(d['value']), d['ms_per_step'], d['config']['avg_iterations'], d.get('last_state_pose',[])[:3])"
grep -a "libliinit_hip" /tmp/e.txt | tail -5
LII_DIAG=1 timeout 600 python bench.py --steps 20000 --long-steps 0 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --workload os1_128_cut3 --map-update 2> /tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('soak cut3 map-update', round(d['value']), d['ms_per_step'], d['config']['avg_iterations'])"
grep -a "libliinit_hip" /tmp/e.txt | tail -4
