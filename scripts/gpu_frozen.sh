#!/bin/bash
# Run a command on the GPU box from a FROZEN copy of the tree, so that the working tree may keep changing while the call waits
# for a slot (gpurun snapshots /root/repo only when it gets a box).  The copy lives under _var/run_<name>/ (git-ignored, travels with
# the snapshot); the command runs inside it with OUT=$GRAFT_REPO_ROOT/gpurun_out/<name> for its outputs.
# usage: bash scripts/gpu_frozen.sh <name> <timeout-seconds> '<command>'
set -o pipefail
NAME=$1; T=$2; shift 2
R=/root/repo
F=$R/_var/run_$NAME
rm -rf "$F"; mkdir -p "$F" "$R/gpurun_out/$NAME"
tar -C "$R" --exclude=./.git --exclude=./gpurun_out --exclude=./_var --exclude=__pycache__ --exclude=./.pytest_cache --exclude='./profiles/round1*' --exclude='./profiles/round2_[a-d]' -cf - . | tar -C "$F" -xf -
bash $R/scripts/gpu.sh "$T" "cd _var/run_$NAME && export OUT=\$GRAFT_REPO_ROOT/gpurun_out/$NAME && mkdir -p \$OUT && $*" > "$R/gpurun_out/$NAME/call.log" 2>&1
rc=$?
rm -rf "$F"
exit $rc
