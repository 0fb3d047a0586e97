#!/bin/bash
# round-5 visit 2: adversarial trajectory test, why the two-half forward differs at B = 64, disturber A/B (LDS-DMA vs plain loads)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05b}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s -k "adversarial" -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > gpurun_out/${TAG}_gan_traj.txt
timeout 400 python tools/dbg/half_vs_full.py full 8000 64 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_half_vs_full.txt
AERO_ALLOW_PACKED_FP32=1 timeout 600 python tools/dbg/disturber_ab.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_disturber_ab.txt
cat gpurun_out/${TAG}_gan_traj.txt gpurun_out/${TAG}_half_vs_full.txt gpurun_out/${TAG}_disturber_ab.txt
