#!/bin/bash
# ablations of the ring conv kernel's K loop (profiling build; outputs are wrong by construction)
cd "$GRAFT_REPO_ROOT" || exit 1
L=gpurun_out/r2f_abl.log
: > $L
for a in 0 256 0 256; do
  echo "== ABL=$a" >> $L
  AERO_RING_ABL=$a AERO_CONV_RING=2 timeout 100 python tools/bench_conv.py --lib aero_amd/libaero_hip_abl.so --layers d0,d1 --iters 20 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
