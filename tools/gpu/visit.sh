#!/bin/bash
# one GPU-box visit: parity tests, bench (+events, +cpu baseline), rocprof kernel trace (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-rX}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o ${TAG} -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.log" 2>&1
echo "rocprof exit $?" >> "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.log"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/${TAG}_prof -type f | head -20
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_bench.log
