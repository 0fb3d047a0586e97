#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for kb in 0 8 16 32 64; do
  echo "== AERO_NORM_CHUNK_KB=$kb"
  AERO_NORM_CHUNK_KB=$kb timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['ms_per_step'], 'apply', k.get('aero_norm_apply_kernel'), 'stats', k.get('aero_norm_stats_kernel'))"
done
