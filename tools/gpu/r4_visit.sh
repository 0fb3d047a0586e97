#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider -k "istft or golden or round_trip" 2>&1 | tail -4 > gpurun_out/r4u_pytest.txt
timeout 300 python tools/dbg/istft_ablation.py 2>&1 | grep "ABL=" > gpurun_out/r4u_istft_abl.txt
timeout 200 python tools/launch_table.py 2>&1 | grep "istft\|stft\|sum of" > gpurun_out/r4u_table.txt
cat gpurun_out/r4u_pytest.txt gpurun_out/r4u_istft_abl.txt gpurun_out/r4u_table.txt
