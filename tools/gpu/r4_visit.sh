#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -6 > gpurun_out/r4n_pytest.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
for i in 1 2 3; do timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done > gpurun_out/r4n_bench.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4n_table.txt
cat gpurun_out/r4n_pytest.txt gpurun_out/r4n_bench.txt; tail -1 gpurun_out/r4n_table.txt
