#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2det}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/dbg/determinism.py --n 50 > gpurun_out/${TAG}_ring.log 2>&1
AERO_CONV_RING=0 timeout 300 python tools/dbg/determinism.py --n 50 --cases conv_stats,conv > gpurun_out/${TAG}_glds8.log 2>&1
timeout 200 python tools/bench_enc0.py > gpurun_out/${TAG}_enc0.log 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_ring.log; grep -v amdgpu.ids gpurun_out/${TAG}_glds8.log; tail -1 gpurun_out/${TAG}_enc0.log
