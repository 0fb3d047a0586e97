#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x -k "pw or full or small or tiny or stress" 2>&1 | tail -6 > gpurun_out/r4f_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_launch_table.txt
AERO_PW=0 timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_launch_table_nopw.txt
cat gpurun_out/r4f_pytest.txt
grep "aero_pw\|sum of" gpurun_out/r4f_launch_table.txt
grep "sum of" gpurun_out/r4f_launch_table_nopw.txt
