cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=lidar_imu_init_amd/lib
for v in old new old new; do
  cp $L/libliinit_hip_$v.so $L/libliinit_hip.so
  timeout 300 python bench.py --no-cpu-baseline --no-pipeline --no-calibration --kernel-profile-steps 160 --long-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v default', round(d['value']), {r['kind']: round(r['avg_us'],2) for r in d['roofline']['kernels']})"
done
