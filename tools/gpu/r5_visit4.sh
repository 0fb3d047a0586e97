#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05d}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/${TAG}_pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest_gpu.log
grep -B25 "short test summary" gpurun_out/${TAG}_pytest_gpu.log | head -60; tail -5 gpurun_out/${TAG}_pytest_gpu.log
for v in 1 0 1 0; do
  AERO_RING_192X128=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench AERO_RING_192X128=$v', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))"
done
