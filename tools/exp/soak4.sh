cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O
for part in voxel index; do
LII_BENCH_ONE_PARTITION=1 LII_BENCH_TRANSPORT=mailbox LII_BENCH_ONE_DEVICE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29504 bench.py --gpus 4 --steps 3000 --map-update --long-steps 0 --no-cpu-baseline --no-calibration --partition $part > $O/soak_$part.json 2> $O/soak_$part.err; echo "$part rc=$?"; grep -a -i "fault\|error" $O/soak_$part.err | head -3
python - <<PY
import json
d=json.loads(open('$O/soak_$part.json').readlines()[-1])
print('$part', round(d['value']), d['config'].get('avg_iterations'), d['config']['transport_why'][:120], d.get('last_state_pose', [])[:3])
PY
done
