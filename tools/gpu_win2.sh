#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
AB_ARGS=--map-update AB_TAG=keep1_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW_KEEP=1
AB_ARGS=--map-update AB_TAG=keep0_ bash tools/ab.sh $1 "tree" "stream100k" LII_WINDOW_KEEP=0
