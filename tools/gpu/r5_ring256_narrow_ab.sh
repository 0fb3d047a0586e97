#!/bin/bash
# 256-row ring tile: 8 waves x 256 steps (shipped) vs 4 waves x 128 steps with a three-slot weight ring, two blocks per CU (AERO_RING_256X128=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 0 1 0 1; do
  echo "AERO_RING_256X128=$v"
  AERO_RING_256X128=$v timeout 200 python tools/launch_table.py 2>&1 | grep "ring_kernel<2, [24], 4, 3\|sum of"
done
AERO_RING_256X128=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_concurrency.py -m gpu -q -x -p no:cacheprovider -k "conv2d or conv_stats or golden or batch64 or two_stream or invarian or determin or forward" --deselect tests/test_gpu_model.py::test_no_aero_switch_is_set_on_the_test_box 2>&1 | tail -3
for v in 0 1 0 1; do
  AERO_RING_256X128=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench AERO_RING_256X128=$v', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))"
done
