#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes (counters only), then
# a kernel trace of the contact-rich regime (scripts/contact_regime.py: every hand lowered onto its box).
# usage: bash scripts/profile_round.sh <name>   -> gpurun_out/<name>/{trace,fetch,write,contact}/...; summarise with
#        python scripts/pmc_summary.py gpurun_out/<name> profiles/<name>
# The program itself follows `--` (python3 ...): no env / bash -c hop under the profiler.
set -o pipefail
NAME=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
S=${DEXSIM_SRC:-$R}   # where bench.py / scripts live (a frozen copy of the tree: scripts/gpu_frozen.sh)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-stagger --no-contact-rich --no-training-like --preroll 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $S/bench.py --steps 200 --warmup 50 $B > $OUT/trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $S/bench.py --steps 20 --warmup 5 $B > $OUT/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $S/bench.py --steps 20 --warmup 5 $B > $OUT/write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/contact -o c -- python3 $S/scripts/contact_regime.py 4096 contact-only > $OUT/contact.log 2>&1
