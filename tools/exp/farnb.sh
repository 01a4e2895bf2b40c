cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python tools/exp/perscan.py 2>&1 | tail -8 | head -3
timeout 300 python bench.py --no-pipeline --no-calibration --kernel-profile-steps 0 --long-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); p=d['parity']; print('default', round(d['value']), d['ms_per_step'], p['dp_max'], p['dtheta_max'], p['iters_equal'], p['effect_num_max_diff'], p['knn_vs_reference_tree']['identical'], p['knn_vs_reference_tree']['of'])"
timeout 900 python -m pytest tests/test_gpu_register.py tests/test_gpu_headline_parity.py tests/test_gpu_map.py tests/test_gpu_end_to_end.py tests/test_gpu_full_size.py -q -x 2>&1 | grep -a -E "passed|failed|Error|^E " | tail -5
