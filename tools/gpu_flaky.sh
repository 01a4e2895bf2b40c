#!/bin/bash
# Flakiness check: the whole -m gpu suite several times on one box, and the map update inside the job under the late-publisher hook at length.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/flaky; mkdir -p $O
for r in 1 2 3; do
  timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $O/pytest_$r.log 2>&1; echo "suite run $r rc=$? $(grep -E 'passed|failed' $O/pytest_$r.log | tail -1)"
done
LII_TEST=emit_late timeout 300 python bench.py --map-update --steps 6000 --no-cpu-baseline --no-pipeline --no-calibration --no-live-traffic --kernel-profile-steps 0 --long-steps 0 > $O/mapupd_late.json 2> $O/mapupd_late.err; echo "map-update under emit_late, 6000 steps rc=$? $(cut -c1-100 $O/mapupd_late.json)"
