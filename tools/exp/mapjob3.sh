cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for t in "" no_gather; do
for f in "" "--map-update-separate"; do
LII_TEST=$t timeout 300 python bench.py --steps 400 --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 --map-update $f 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('map-update [$t] [$f]', round(d['value']), d['ms_per_step'])"
done
LII_TEST=$t timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-calibration --kernel-profile-steps 0 --long-steps 0 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); p=d['complete_pipeline']; print('pipeline [$t]', round(p['value']), round(p['pageable_source']['value']), round(p['serial_upload']['value']))"
done
