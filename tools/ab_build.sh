#!/bin/bash
# Builds libliinit_hip.so as an experiment variant for tools/ab.sh (A/B measurements on the GPU box).
#   tools/ab_build.sh <name> [git revision | -] [extra compiler flags ...]
# <name>: the variant lands in build_ab/<name>/libliinit_hip.so (git-ignored, travels with gpurun);  revision: build that
# revision's sources (a detached worktree under /tmp) instead of the working tree ("-": the working tree).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:--}; shift; shift || true
SRC=$ROOT
if [ "$REV" != "-" ]; then
  SRC=/tmp/lii_ab_wt_$NAME
  rm -rf "$SRC"; git -C "$ROOT" worktree prune; git -C "$ROOT" worktree add --detach "$SRC" "$REV" > /dev/null 2>&1
fi
mkdir -p "$ROOT/build_ab/$NAME"
make -C "$SRC/lidar_imu_init_amd/csrc" -j8 OUT="$ROOT/build_ab/$NAME" OBJ="/tmp/lii_ab_obj_$NAME" EXTRA="$*" 2>&1 | grep -E "error|warning: (?!argument unused)" || true
ls -la "$ROOT/build_ab/$NAME/libliinit_hip.so"
if [ "$REV" != "-" ]; then git -C "$ROOT" worktree remove --force "$SRC"; fi
