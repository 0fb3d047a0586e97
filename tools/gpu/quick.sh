#!/bin/bash
# quick check: GPU parity subset + bench line
cd "$GRAFT_REPO_ROOT" || exit 1
K=${1:-"conv or norm"}
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "$K" -p no:cacheprovider 2>&1 | tail -2
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/quick_bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_bench.log').read())
print('ms_per_step', d['ms_per_step'])
for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:14]: print(f'  {k:45s} {v:.3f}')
PY
