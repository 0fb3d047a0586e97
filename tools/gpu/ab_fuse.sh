#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_stats" -p no:cacheprovider 2>&1 | tail -1
for cfg in "AERO_FUSE_DCONV=0 AERO_FUSE_STATS=0" "AERO_FUSE_DCONV=1 AERO_FUSE_STATS=0" "AERO_FUSE_DCONV=0 AERO_FUSE_STATS=1" "AERO_FUSE_DCONV=1 AERO_FUSE_STATS=1"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['ms_per_step'], 'norm_apply', k.get('aero_norm_apply_kernel'), 'stats', k.get('aero_norm_stats_kernel'))"
  env $cfg timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
done
