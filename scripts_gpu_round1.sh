#!/bin/bash
# one GPU-box visit: parity tests, smoke, bench, rocprof kernel trace (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
lscpu | head -20 >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1
echo "rocprof exit $?" >> "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof -name "*.csv" | head -20
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
