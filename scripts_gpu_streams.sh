#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-stX}
mkdir -p gpurun_out
for ns in 1 2 4; do echo "== AERO_STREAMS=$ns" >> gpurun_out/${TAG}_bench.log; AERO_CONV_MODE=1 AERO_STREAMS=$ns timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events >> gpurun_out/${TAG}_bench.log 2>&1; done
grep -E "^==|^\{" gpurun_out/${TAG}_bench.log | cut -c1-330
