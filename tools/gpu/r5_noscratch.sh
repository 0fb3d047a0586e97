#!/bin/bash
# inference kernels without a private segment (LSTM ring: 32-bit bases; 8-wave conv tile with statistics: pinned sums)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
  echo "== pass $rep"
  timeout 200 python tools/launch_table.py 2>&1 | grep "glds8_kernel<3, 32, true\|lstm_ring\|sum of"
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for v in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))"
done
