cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fin5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/driver_form.json').readline()); r=d['roofline']; print(round(d['value']), round(d['value_long']['value']), d['slowest_step']['ms'], r['frac'], r['traffic'], r['traffic_source'][:40], round(d['complete_pipeline']['value']), d['cpu_baseline']['value'])"
