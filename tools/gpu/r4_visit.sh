#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -4 > gpurun_out/r4o_pytest.txt
AERO_NORM_FAST=0 timeout 200 python tools/launch_table.py 2>&1 | grep "norm_apply\|sum of" > gpurun_out/r4o_norm_old.txt
timeout 200 python tools/launch_table.py 2>&1 | grep "norm_apply\|sum of" > gpurun_out/r4o_norm_fast.txt
cat gpurun_out/r4o_pytest.txt; paste -d'|' <(cut -c1-60 gpurun_out/r4o_norm_old.txt) <(cut -c1-90 gpurun_out/r4o_norm_fast.txt)
