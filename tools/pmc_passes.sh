#!/bin/bash
# Hardware-counter passes over tools/pmc_spmv.py (each pass its own run); results under
# gpurun_out/pmc_<tag>/<pass>/.  Usage: tools/pmc_passes.sh <tag>   (LSK_* env selects the kernel form)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-default}
cd /tmp && export TMPDIR=/tmp
run_pass() {
  local name=$1; shift
  local out=$R/gpurun_out/pmc_$TAG/$name
  mkdir -p "$out"
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out" -- python3 "$R/tools/pmc_spmv.py" > "$out/log.txt" 2>&1 || { echo "pass $name failed"; tail -5 "$out/log.txt"; return 1; }
}
run_pass p1 SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
run_pass p2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_REQUEST_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum &&
run_pass p3 TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU
