#!/bin/bash
# round-5 visit 3: full GPU suite on the tree with the fused last layer and the batch-invariant STFT statistics; launch table; bench
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05c}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/${TAG}_pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 400 python tools/dbg/half_vs_full.py full 8000 64 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_half_vs_full.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
AERO_FUSE_TAIL=0 timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/${TAG}_launch_table_unfused.txt
timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err
AERO_FUSE_TAIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events > gpurun_out/${TAG}_bench_unfused.log 2>&1
grep -B30 "short test summary" gpurun_out/${TAG}_pytest_gpu.log | head -80; tail -6 gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_half_vs_full.txt
tail -8 gpurun_out/${TAG}_launch_table.txt; echo unfused; cat gpurun_out/${TAG}_launch_table_unfused.txt
for f in gpurun_out/${TAG}_bench.log gpurun_out/${TAG}_bench_unfused.log; do grep '^{' $f | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('ms_per_step_one_at_a_time'), r.get('frac'), r.get('conv_stack'), r.get('step_mfma_frac'), d['config'].get('other_configs'))
"; done
