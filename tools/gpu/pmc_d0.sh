#!/bin/bash
# PMC: LDS vs MFMA utilisation of the dominant conv kernel (separate passes; no trace domains with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pd}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n.log" 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run 2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run 3 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
cd "$GRAFT_REPO_ROOT"; ls gpurun_out/${TAG}_pmc1 | head -2
