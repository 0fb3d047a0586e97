#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -8 > gpurun_out/r4s_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4s_table.txt
for i in 1 2; do timeout 200 python bench.py --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])" >> gpurun_out/r4s_bench.txt; done
cat gpurun_out/r4s_pytest.txt; grep "pw_kernel\|sum of" gpurun_out/r4s_table.txt | cut -c1-130; cat gpurun_out/r4s_bench.txt
