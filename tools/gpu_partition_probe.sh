#!/bin/bash
# Read-only look at the box's partition state (the pool refuses every command that would change it: profiles/r06_partition_probe.md).
# usage: bash tools/gpu_partition_probe.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O; L=$O/probe.log
run() { echo "=== \$ $*" >> $L; timeout 60 "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
run id; run ls -la /dev/kfd /dev/dri
run rocm-smi --showcomputepartition --showmemorypartition
run amd-smi static --partition
timeout 120 python -c "import torch; print('torch devices:', torch.cuda.device_count()); [print(i, torch.cuda.get_device_properties(i).name, torch.cuda.get_device_properties(i).multi_processor_count, torch.cuda.get_device_properties(i).total_memory >> 30, 'GiB') for i in range(torch.cuda.device_count())]" >> $L 2>&1
tail -30 $L
