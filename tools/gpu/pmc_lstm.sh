#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-plX}
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_lstm.py > gpurun_out/${TAG}_lstmbench.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1" -o p1 -- python "$GRAFT_REPO_ROOT/tools/bench_lstm.py" --iters 2 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2" -o p2 -- python "$GRAFT_REPO_ROOT/tools/bench_lstm.py" --iters 2 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2.log" 2>&1
cd "$GRAFT_REPO_ROOT"; cat gpurun_out/${TAG}_lstmbench.log; tail -3 gpurun_out/${TAG}_pmc2.log
