#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pw}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_pw.log
: > $L
timeout 120 python tools/bench_conv.py --layers pw768,pw384,pw96,pw192,pw768k,d0,d1,d2,d3 --iters 20 2>&1 | grep -v amdgpu.ids >> $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv or dconv" -p no:cacheprovider 2>&1 | tail -2 >> $L
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> $L
cat $L
