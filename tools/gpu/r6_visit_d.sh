#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r06d_pytest.log
cat gpurun_out/r06d_pytest.log
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d_launch_table.txt
grep "lstm\|sum of" gpurun_out/r06d_launch_table.txt
AERO_LSTM_FRAME_MAJOR=0 timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d_launch_table_frame_minor.txt
grep "lstm\|sum of" gpurun_out/r06d_launch_table_frame_minor.txt
for i in 1 2; do timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep '^K=' | cut -c1-60; done | tee gpurun_out/r06d_pipeline.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline 2>/dev/null | cut -c1-300
