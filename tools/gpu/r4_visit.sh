#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -4 > gpurun_out/r4p_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4p_table.txt
cat gpurun_out/r4p_pytest.txt; grep "squeeze\|skinny\|norm_stats\|sum of" gpurun_out/r4p_table.txt | cut -c1-150
