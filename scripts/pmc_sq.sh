#!/bin/bash
# SQ counter passes for the sub-step kernel (each pass its own process; counters only, no API traces).
# usage (on the GPU box): bash scripts/pmc_sq.sh   -> gpurun_out/sq_*/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1
run() { # name, counters...
  n=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/sq_$n -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sq_$n.log 2>&1
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU &&
run c SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8
