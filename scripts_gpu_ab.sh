#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-abX}
mkdir -p gpurun_out
for m in 1 2 3; do echo "== AERO_CONV_MODE=$m" >> gpurun_out/${TAG}_convbench.log; AERO_CONV_MODE=$m python tools/bench_conv.py --iters 20 >> gpurun_out/${TAG}_convbench.log 2>&1; done
timeout 600 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
for m in 0 1 2; do echo "== AERO_CONV_MODE=$m" >> gpurun_out/${TAG}_bench.log; AERO_CONV_MODE=$m timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> gpurun_out/${TAG}_bench.log 2>&1; done
cat gpurun_out/${TAG}_convbench.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log; grep -E "^==|^\{" gpurun_out/${TAG}_bench.log | cut -c1-330
