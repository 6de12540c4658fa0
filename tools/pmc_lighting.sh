#!/bin/bash
# PMC passes over the fused lighting kernel (scene G-buffer, 4K). Counters only with --kernel-trace (no API tracing).
#   bash tools/pmc_lighting.sh gpurun_out/pmc_r2
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { # name counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out/$name" -- python3 "$R/tools/bench_kernels.py" --gbuffer scene --iters 10 > "$R/$out/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run ta TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
