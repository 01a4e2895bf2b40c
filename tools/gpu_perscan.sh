#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "--- tree"; LII_DIAG=1 timeout 200 python tools/perscan.py 2>&1 | grep -v amdgpu | grep "^[0-7] us\|parked\|pre-armed" | cut -c1-90
echo "--- c32"; LII_LIB=$PWD/build_ab/c32/libliinit_hip.so LII_DIAG=1 timeout 200 python tools/perscan.py 2>&1 | grep -v amdgpu | grep "^[0-7] us\|parked" | cut -c1-90
