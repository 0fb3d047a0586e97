#!/bin/bash
# whole-model PMC passes: instruction counts and busy/wait cycles per kernel (no trace domains combined with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pmX}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n.log" 2>&1; }
run 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA
run 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/${TAG}_pmc1/*counter_collection.csv gpurun_out/${TAG}_pmc2/*counter_collection.csv
