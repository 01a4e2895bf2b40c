#!/bin/bash
# full -m gpu suite on the current library, then A/B against lib .prev (same box), then the timeline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
L=lidar_imu_init_amd/lib
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log | cut -c1-200
cp $L/libliinit_hip.so $L/libliinit_hip.so.new
for rep in 1 2; do for which in prev new; do
  cp $L/libliinit_hip.so.$which $L/libliinit_hip.so
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-pipeline > $O/bench_$which.json 2> $O/bench_$which.err
  python -c "
import json; d=json.loads(open('$O/bench_$which.json').readline()); print('$which', round(d['value']), round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
cp $L/libliinit_hip.so.new $L/libliinit_hip.so
bash tools/gpu_prof.sh $1 cur 2>&1 | tail -22
