#!/bin/bash
# whole-model PMC passes: instruction counts and busy/wait cycles per kernel (no trace domains combined with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pmX}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/${TAG}_bench.log
cd /tmp
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n.log" 2>&1; }
run 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA
run 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
cd "$GRAFT_REPO_ROOT"; cat gpurun_out/${TAG}_bench.log; ls gpurun_out/${TAG}_pmc1 | head -3
