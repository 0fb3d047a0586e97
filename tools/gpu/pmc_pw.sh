#!/bin/bash
# PMC profile of a pointwise conv (separate counter passes; no trace domains combined with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pwX}
LAY=${2:-pw768}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers $LAY --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n.log" 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run 2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
run 3 TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_WRREQ_STALL_sum
run 4 SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
run 5 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
cd "$GRAFT_REPO_ROOT"; tail -2 gpurun_out/${TAG}_pmc1.log | cut -c1-200; ls gpurun_out/${TAG}_pmc*/ 2>/dev/null | head
