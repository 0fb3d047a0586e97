#!/bin/bash
# iSTFT second form, DFT-form STFT with prefetched spans, strided conv on the 192-row ring tile, loss-trajectory test, full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 2>&1 | tail -25 > gpurun_out/r4h_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4h_launch_table.txt
AERO_ISTFT_V2=0 timeout 200 python tools/launch_table.py 2>&1 | grep "istft\|sum of" > gpurun_out/r4h_istft_v1.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
for i in 1 2; do timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done > gpurun_out/r4h_bench.txt
cat gpurun_out/r4h_pytest.txt | tail -12
grep "stft\|ring_kernel<2, 4, 3, 1\|M=384 C=192+0 taps=8\|sum of" gpurun_out/r4h_launch_table.txt; cat gpurun_out/r4h_istft_v1.txt gpurun_out/r4h_bench.txt
