#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -3 > gpurun_out/r4r_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4r_table.txt
cat gpurun_out/r4r_pytest.txt; grep "norm_stats\|sum of" gpurun_out/r4r_table.txt | cut -c1-120
