#!/bin/bash
# round-4 fifth visit: the streaming pointwise kernel (k_pw.h) in the model, the FFT part without packed-fp32 (fence), A/B benches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r4e_pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
timeout 200 $B > gpurun_out/r4e_bench_default.log 2>&1
AERO_PW=0 timeout 200 $B > gpurun_out/r4e_bench_nopw.log 2>&1
AERO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_nopk_all.so timeout 200 $B > gpurun_out/r4e_bench_nopk_all.log 2>&1
timeout 200 $B > gpurun_out/r4e_bench_default2.log 2>&1
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4e_launch_table.txt
cat gpurun_out/r4e_pytest.txt
for f in default nopw nopk_all default2; do echo $f; grep '^{' gpurun_out/r4e_bench_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
grep "aero_pw\|sum of" gpurun_out/r4e_launch_table.txt
