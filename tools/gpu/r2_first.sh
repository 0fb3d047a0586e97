#!/bin/bash
# round-2 first visit: all GPU tests (new: stress parity, config 4 full size, predict path, RCCL), baseline bench of this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider -s > gpurun_out/r2a_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2a_pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a_bench.log 2>&1
grep -E "rel-L2|passed|failed|FAILED|Error" gpurun_out/r2a_pytest_gpu.log | tail -40
tail -1 gpurun_out/r2a_bench.log | cut -c1-400
