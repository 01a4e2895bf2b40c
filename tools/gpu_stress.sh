#!/bin/bash
# result hand-off / pre-armed prologue under repetition: tools/stress_result.py, then the whole GPU suite twice
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
LII_DIAG=1 timeout 600 python tools/stress_result.py ${1:-30000} 2>&1 | grep -v amdgpu.ids | tail -6
for r in 1 2; do timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -2; done
