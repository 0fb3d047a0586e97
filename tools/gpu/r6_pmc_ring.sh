#!/bin/bash
# round 6: what bounds the decoder 3x3 ring tiles -- effective clock, MFMA-pipe busy, LDS, texture path; separate --pmc passes, no trace domains
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r06
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_conv.py --layers d0,d1,d2,d3 --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ring_random.txt
python tools/bench_conv.py --layers d0,d1,d2,d3 --iters 20 --zeros 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ring_zeros.txt
cd /tmp
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_ringpmc$n" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d1,d2,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_ringpmc$n.log" 2>&1; }
run 1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS
run 2 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run 3 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run 4 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
run 5 GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
cd "$GRAFT_REPO_ROOT"
python tools/pmc_ring_table.py $(find gpurun_out/${TAG}_ringpmc* -name "*counter_collection.csv") --json gpurun_out/${TAG}_pmc_clock.json > gpurun_out/${TAG}_pmc_ring_table.txt 2>&1
cat gpurun_out/${TAG}_ring_random.txt gpurun_out/${TAG}_ring_zeros.txt gpurun_out/${TAG}_pmc_ring_table.txt
head -3 $(find gpurun_out/${TAG}_ringpmc1 -name "*counter_collection.csv" | head -1)
tail -3 gpurun_out/${TAG}_ringpmc4.log gpurun_out/${TAG}_ringpmc5.log
rm -rf gpurun_out/${TAG}_ringpmc*/
