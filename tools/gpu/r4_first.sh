#!/bin/bash
# round-4 first visit: GPU suite on the part-built library (+ the new concurrency fence), the iSTFT experiments, train.py on one GPU,
# a bench line and the launch table of this box as the baseline of the round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/dbg/istft_probe.py > gpurun_out/r4a_istft_probe.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/r4a_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4a_pytest_gpu.log
timeout 300 python train.py experiment=aero_11-44_512_256 experiment.batch_size=2 steps=4 > gpurun_out/r4a_train_py.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/r4a_bench.log 2>gpurun_out/r4a_bench.err
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a_launch_table.txt
grep -v "Warn\|warn" gpurun_out/r4a_istft_probe.txt | tail -20
tail -15 gpurun_out/r4a_pytest_gpu.log
tail -8 gpurun_out/r4a_train_py.txt
grep '^{' gpurun_out/r4a_bench.log | cut -c1-400
tail -2 gpurun_out/r4a_launch_table.txt
