#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-extra-configs --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: ms/step', d['ms_per_step'], 'one at a time', d['ms_per_step_one_at_a_time'])"; }
{
AERO_PIPELINE=3 run "default"
GPU_MAX_HW_QUEUES=8 AERO_PIPELINE=3 run "GPU_MAX_HW_QUEUES=8"
AERO_PIPELINE=4 run "depth 4"
GPU_MAX_HW_QUEUES=8 AERO_PIPELINE=4 run "GPU_MAX_HW_QUEUES=8 depth 4"
AERO_PIPELINE=3 run "default"
GPU_MAX_HW_QUEUES=8 AERO_PIPELINE=6 run "GPU_MAX_HW_QUEUES=8 depth 6"
} > gpurun_out/r4z.txt
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "pipeline or two_stream or predict" 2>&1 | tail -2 >> gpurun_out/r4z.txt
cat gpurun_out/r4z.txt
