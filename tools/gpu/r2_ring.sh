#!/bin/bash
# ring conv kernel: parity (op tests), race screen, A/B against the k_conv.h kernels, whole-model bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2b}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_ring.log
: > $L
python tools/dbg/batch_invariance.py wide 24000 4 > gpurun_out/dbg_binv_wide.log 2>&1
python tools/dbg/batch_invariance.py full 8000 3 > gpurun_out/dbg_binv_full.log 2>&1
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k conv2d -p no:cacheprovider >> $L 2>&1
for cfg in "AERO_CONV_RING=0" "AERO_CONV_RING=2"; do
  echo "== $cfg" >> $L
  env $cfg timeout 200 python tools/bench_conv.py --layers d0,d1,d2,d3 --iters 30 --race 10 >> $L 2>&1
done
for cfg in "AERO_CONV_RING=0" "AERO_CONV_RING=1" "AERO_CONV_RING=2"; do
  echo "== bench $cfg" >> $L
  env $cfg timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | cut -c1-330 >> $L
done
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "rel-L2|passed|failed" >> $L
cat $L
echo; tail -14 gpurun_out/dbg_binv_wide.log; tail -14 gpurun_out/dbg_binv_full.log
