#!/bin/bash
# round 6 visit C: full GPU suite on the pruned library (+ the schedule-equality test), family marginals and dispatch trace of the default
# serving loop, the one-ring-block-per-CU experiment, the training step's graph under the runtime's graph knobs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r06c_pytest_gpu.log
tail -3 gpurun_out/r06c_pytest_gpu.log
timeout 600 python tools/dbg/pipeline_ablation.py 30 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/r06c_pipeline_ablation.txt
cat gpurun_out/r06c_pipeline_ablation.txt
for v in 0 1 0 1; do
  echo "AERO_RING_ONE_PER_CU=$v: $(AERO_RING_ONE_PER_CU=$v timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep '^K=' | cut -c1-28 | tr '\n' ' ') | d2,d3 alone: $(AERO_RING_ONE_PER_CU=$v python tools/bench_conv.py --layers d2,d3 --iters 20 2>&1 | grep -v amdgpu | cut -c1-22 | tr '\n' ' ')"
done | tee gpurun_out/r06c_ring_one_per_cu.txt
cd /tmp
TRACE=plain timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r06c_trace" -o t -- python "$GRAFT_REPO_ROOT/tools/dbg/sched_sweep.py" 30 2>&1 | grep "ms per batch"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/r06c_trace" -name "*kernel_trace.csv" | head -1)
python "$GRAFT_REPO_ROOT/tools/dbg/trace_overlap.py" "$f" "default serving loop (waits)" > "$GRAFT_REPO_ROOT/gpurun_out/r06c_trace_overlap_default.txt"
cat "$GRAFT_REPO_ROOT/gpurun_out/r06c_trace_overlap_default.txt"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r06c_trace"
cd "$GRAFT_REPO_ROOT"
tools/gpu/r6_train_graph_env.sh
