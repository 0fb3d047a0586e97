#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/launch_table.py 2>&1 | grep "ring_kernel\|sum of" > gpurun_out/r4j_ring_default.txt
AERO_RING_HALF=1 timeout 200 python tools/launch_table.py 2>&1 | grep "ring_kernel\|sum of" > gpurun_out/r4j_ring_half.txt
T="tests/test_gpu_model.py::test_two_stream_forward_equals_one_stream tests/test_gpu_model.py::test_full_model_batch64_matches_golden_and_is_batch_invariant tests/test_gpu_model.py::test_full_model_golden tests/test_gpu_determinism.py tests/test_gpu_distrib.py"
AERO_RING_HALF=1 timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r4j_pytest_half.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
for e in AERO_RING_HALF=0 AERO_RING_HALF=1 AERO_RING_HALF=0 AERO_RING_HALF=1; do echo -n "$e "; env $e timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > gpurun_out/r4j_bench.txt
cat gpurun_out/r4j_ring_default.txt gpurun_out/r4j_ring_half.txt gpurun_out/r4j_pytest_half.txt gpurun_out/r4j_bench.txt
