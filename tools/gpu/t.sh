#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k conv_stats -p no:cacheprovider 2>&1 | tail -2
echo "== fused"; AERO_FUSE_STATS=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "FAILED|passed|failed" | head
AERO_FUSE_STATS=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | grep -o '"ms_per_step": [0-9.]*'
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | grep -o '"ms_per_step": [0-9.]*'
