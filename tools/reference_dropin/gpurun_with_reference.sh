#!/bin/bash
# Runs a command on the MI355X box with the reference package shipped beside the repo as
# git-ignored scratch (_reference_scratch/, removed again afterwards -- never committed).
#   tools/reference_dropin/gpurun_with_reference.sh [--timeout S] -- '<command>'
# On the box TN_REFERENCE_DIR=$GRAFT_REPO_ROOT/_reference_scratch is what the harness and
# tests/test_gpu_reference_dropin.py look for.
set -u
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
SRC="${TN_REFERENCE_SRC:-/root/reference}"
SCRATCH="$REPO/_reference_scratch"
rm -rf "$SCRATCH"; mkdir -p "$SCRATCH"
cp -r "$SRC/tensornetwork" "$SCRATCH/tensornetwork"
find "$SCRATCH" -name "__pycache__" -type d -prune -exec rm -rf {} +
trap 'rm -rf "$SCRATCH"' EXIT
/usr/local/graft/bin/gpurun "$@"
