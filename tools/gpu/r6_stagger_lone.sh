#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 0 3 4 0 2 5; do
  echo "AERO_STAGGER=$v: $(AERO_STAGGER=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pipelined', d['ms_per_step'], 'one at a time', d['ms_per_step_one_at_a_time'])")"
done | tee gpurun_out/r06_stagger_lone_forward.txt
