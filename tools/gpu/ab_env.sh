#!/bin/bash
# A/B of one environment switch on the bench line:  ab_env.sh VAR v1 v2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print(d['ms_per_step'], {a:b for a,b in sorted(k.items(), key=lambda kv:-kv[1])[:9]})"
done
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
