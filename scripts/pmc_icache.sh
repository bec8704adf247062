#!/bin/bash
# Instruction-cache counters of the step kernel (own pass, counters only).  usage: bash scripts/pmc_icache.sh <name>
# -> gpurun_out/<name>/icache/  (k_physics4 is ~420 KB of straight-line code against a 64 KB instruction cache per CU pair)
set -o pipefail
NAME=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-stagger --no-contact-rich --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/icache -o p -- python3 $R/bench.py $B > $OUT/icache.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/ifetch -o p -- python3 $R/bench.py $B > $OUT/ifetch.log 2>&1
