#!/bin/bash
# PMC profile of the dominant conv kernel (separate counter passes; no trace domains combined with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pX}
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_conv.py --iters 20 > gpurun_out/${TAG}_convbench.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1" -o p1 -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2" -o p2 -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc3" -o p3 -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc3.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc4" -o p4 -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc4.log" 2>&1
cd "$GRAFT_REPO_ROOT"; cat gpurun_out/${TAG}_convbench.log; find gpurun_out/${TAG}_pmc* -name "*.csv" | head; tail -2 gpurun_out/${TAG}_pmc1.log
