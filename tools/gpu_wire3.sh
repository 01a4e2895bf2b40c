#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_replay_host.py tests/test_gpu_scan_ops.py tests/test_gpu_end_to_end.py -m gpu -q --timeout 600 2>&1 | tail -4
bash tools/gpu_wire.sh $1
