#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL version" | tail -4 > gpurun_out/r4q_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4q_table.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
for e in AERO_STREAMS=0 AERO_STREAMS=1 AERO_STREAMS=0 AERO_STREAMS=3; do echo -n "$e "; env $e timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > gpurun_out/r4q_bench.txt
timeout 200 python tools/profile_train.py 2 2>&1 | grep -v "amdgpu.ids\|Warn" | head -30 > gpurun_out/r4q_train_profile_b2.txt
cat gpurun_out/r4q_pytest.txt; grep "norm_stats\|pw M=48 C=96 F=64\|sum of" gpurun_out/r4q_table.txt | cut -c1-140; cat gpurun_out/r4q_bench.txt; head -14 gpurun_out/r4q_train_profile_b2.txt
