#!/bin/bash
# round-5 visit 1: the GPU parity suite on the tree with the ADVICE / hygiene fixes, the new third-party-victim fences and the attention
# block skip; why the two-half-batch forward differs at B = 64 (tools/dbg/half_vs_full.py); launch table with / without the skip; bench
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/${TAG}_pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python tools/dbg/half_vs_full.py full 8000 64 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_half_vs_full.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
AERO_ATTN_SKIP=0 timeout 200 python tools/launch_table.py 2>&1 | grep "attn" > gpurun_out/${TAG}_launch_table_noskip.txt
timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err
tail -6 gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_half_vs_full.txt; grep attn gpurun_out/${TAG}_launch_table.txt; cat gpurun_out/${TAG}_launch_table_noskip.txt
tail -1 gpurun_out/${TAG}_launch_table.txt
grep '^{' gpurun_out/${TAG}_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('ms_per_step_one_at_a_time'), d['roofline'].get('conv_stack'), d['roofline'].get('stft'), d['roofline'].get('istft'), d['roofline'].get('step_mfma_frac'), d['config'].get('other_configs'))
"
