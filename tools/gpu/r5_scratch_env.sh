#!/bin/bash
# kernels with a private segment (spilled registers): does the runtime's scratch management cost the launch ~40 us?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
for v in default noreclaim noasync; do
  unset HSA_NO_SCRATCH_RECLAIM HSA_ENABLE_SCRATCH_ASYNC_RECLAIM
  [ $v = noreclaim ] && export HSA_NO_SCRATCH_RECLAIM=1
  [ $v = noasync ] && export HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
  echo "== $v (pass $rep)"
  timeout 200 python tools/launch_table.py 2>&1 | grep "glds8_kernel<3, 32, true\|lstm_ring\|sum of"
done
done
for v in default noreclaim noasync default noreclaim noasync; do
  unset HSA_NO_SCRATCH_RECLAIM HSA_ENABLE_SCRATCH_ASYNC_RECLAIM
  [ $v = noreclaim ] && export HSA_NO_SCRATCH_RECLAIM=1
  [ $v = noasync ] && export HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench $v', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))"
done
