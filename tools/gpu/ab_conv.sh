#!/bin/bash
# A/B of conv tile variants on the decoder 3x3 shapes (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ab}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_ab.log
: > $L
for cfg in "AERO_CONV_BM256=0" "AERO_CONV_BM256=1" "AERO_CONV_BM256=2"; do
  echo "== $cfg" >> $L
  env $cfg timeout 120 python tools/bench_conv.py --layers d0,d1,d2,d3 --iters 30 >> $L 2>&1
done
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k conv -p no:cacheprovider >> $L 2>&1
for cfg in "AERO_CONV_BM256=0" "AERO_CONV_BM256=1" "AERO_CONV_BM256=2"; do
  echo "== bench $cfg" >> $L
  env $cfg timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events >> $L 2>&1
done
cat $L
