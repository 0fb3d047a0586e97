#!/bin/bash
# quick A/B visit: conv micro-benchmark (+ optional env), GPU parity tests, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-qX}
mkdir -p gpurun_out
python tools/bench_conv.py --iters 20 > gpurun_out/${TAG}_convbench.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
cat gpurun_out/${TAG}_convbench.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
