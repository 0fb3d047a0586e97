#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do for v in 0 1; do
  echo "AERO_BWD_ARENA=$v: $(AERO_BWD_ARENA=$v timeout 300 python bench.py --extra-configs-worker --steps 12 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('['):
        r = json.loads(l)
        print('train eager', r[1].get('ms_per_step'), 'graph', r[1].get('ms_per_step_hip_graph'), '| adversarial', r[2].get('ms_per_step'))
")"
done; done | tee gpurun_out/r06_train_arena_ab.txt
