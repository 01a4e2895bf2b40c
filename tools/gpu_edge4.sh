#!/bin/bash
# Round 6: the --edge / --edge-wide records (every scan looks past the map's edge) + the whole GPU suite of the tree.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; grep -a -E "passed|failed|error|FAILED|ERROR|^E " $O/pytest_full.txt | tail -25 > $O/pytest.txt; cat $O/pytest.txt
for e in edge edge-wide; do
timeout 400 python bench.py --$e --no-pipeline --no-calibration --no-live-traffic > $O/bench_$e.json 2> $O/bench_$e.err
python -c "
import json; d=json.loads(open('$O/bench_$e.json').readline()); print('$e', round(d['value']), d['ms_per_step'], d['edge'], d.get('parity'))"
done
