#!/bin/bash
# what bounds the ring kernel's K loop for the 192-row tile (D3: K = 864) and the 256-row tile (D0)?  ablation builds (results WRONG by design)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_ringabl.so
{
for abl in 0 1 2 4 6 8 32 64 128; do
  echo "AERO_RING_ABL=$abl"; AERO_RING_ABL=$abl timeout 100 python tools/bench_conv.py --layers d0,d2,d3 --iters 10 --lib $L 2>&1 | grep "^d"
done
} > gpurun_out/r4k_ring_ablation.txt 2>&1
cat gpurun_out/r4k_ring_ablation.txt
