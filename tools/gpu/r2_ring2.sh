#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2d}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_ring.log
: > $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k conv2d -p no:cacheprovider 2>&1 | tail -3 >> $L
for cfg in "AERO_CONV_RING=0" "AERO_CONV_RING=2"; do
  echo "== $cfg" >> $L
  env $cfg timeout 200 python tools/bench_conv.py --layers d0,d1,d2,d3 --iters 30 --race 10 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
