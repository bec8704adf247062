#!/bin/bash
# Instruction-issue counters of the step kernel (own passes, counters only): headline regime (bench.py) and contact-rich regime
# (scripts/contact_regime.py contact-only).  usage: bash scripts/pmc_issue.sh <name> -> gpurun_out/<name>/issue_{h,c}{1,2}/
set -o pipefail
NAME=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
S=${DEXSIM_SRC:-$R}   # where bench.py / scripts live (a frozen copy of the tree: scripts/gpu_frozen.sh)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-stagger --no-contact-rich --no-training-like --preroll 0 --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/issue_h1 -o p -- python3 $S/bench.py $B > $OUT/issue_h1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/issue_h2 -o p -- python3 $S/bench.py $B > $OUT/issue_h2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/issue_c1 -o p -- python3 $S/scripts/contact_regime.py 4096 contact-only > $OUT/issue_c1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/issue_c2 -o p -- python3 $S/scripts/contact_regime.py 4096 contact-only > $OUT/issue_c2.log 2>&1
