#!/bin/bash
# config 4 (T = 376): the 256-row ring tile over 128 steps (rule: odd number of 128-step tiles) vs over 256 steps (AERO_RING_256X128=0)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 0 2 0 2; do
  echo "AERO_RING_256X128=$v"
  AERO_RING_256X128=$v timeout 200 python tools/launch_table.py --config4 2>&1 | grep "ring_kernel<2, [24], 4, 3\|sum of"
done
timeout 200 python tools/launch_table.py --config4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_launch_table_config4.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for v in 0 2 0 2; do
  AERO_RING_256X128=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench AERO_RING_256X128=$v', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'), d['config']['other_configs']['config4_inference'])"
done
