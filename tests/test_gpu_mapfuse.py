"""The map update in four launches (round 6: the fold's hash insert rides in k_map_decide, the inserts' cells are found / created inside the fold
launch - beside the fold's own cell lookups) against the six-launch form of rounds 2 - 5 (LII_MAP_FUSE=0) on a map that GROWS under the updates:
tools/stress_mapfuse.py registers the same stream on two handles and compares the maps as point SETS (KD_TREE::Add_Points leaves a set,
include/ikd-Tree/ikd_Tree.cpp:381-456) and the registered states bit for bit.  Also under LII_TEST=pred_small: every update is enqueued for list
sizes that are too small, finds its lists outgrown and is repeated with the exact ones - the table the decision launch had filled is cleared first."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hook", ["", "pred_small", "map_tight"])
def test_fused_and_six_launch_forms_leave_the_same_map(hook):
    env = dict(os.environ)
    env.pop("LII_MAP_FUSE", None)
    if hook:
        env["LII_TEST"] = hook
    else:
        env.pop("LII_TEST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_mapfuse.py"), "30", "vlp16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "identical sets and states" in r.stdout
