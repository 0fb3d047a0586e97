#!/bin/bash
# r2_quick.sh TAG script.py [pytest -k expr]: one micro-benchmark + selected GPU op tests
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; SCRIPT=$2; K=$3
mkdir -p gpurun_out
if [ -n "$K" ]; then timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "$K" 2>&1 | tail -3; fi
timeout 200 python $SCRIPT 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_quick.log
