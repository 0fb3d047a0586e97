#!/bin/bash
# round-4 second visit: dissect the FFT-form STFT failure next to the ring tile; window-bisect the iSTFT-vs-forward finding; re-run the fence
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/dbg/stft_dissect.py 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/r4b_stft_dissect.txt
timeout 300 python tools/dbg/istft_window.py 6 50 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/r4b_istft_window.txt
timeout 300 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r4b_pytest.txt
cat gpurun_out/r4b_stft_dissect.txt; cat gpurun_out/r4b_istft_window.txt; tail -25 gpurun_out/r4b_pytest.txt
