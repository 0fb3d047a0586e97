#!/bin/bash
# round 6 visit A: fused STFT + pitched iSTFT hand-off -- targeted parity tests, launch table, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r06a_pytest.log
cat gpurun_out/r06a_pytest.log
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06a_launch_table.txt
head -3 gpurun_out/r06a_launch_table.txt; tail -4 gpurun_out/r06a_launch_table.txt
AERO_STFT_FUSED=0 AERO_PITCHED_OUT=0 timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06a_launch_table_before.txt
head -3 gpurun_out/r06a_launch_table_before.txt; tail -4 gpurun_out/r06a_launch_table_before.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06a_bench.json') if l.startswith('{')][0])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_one_at_a_time')})
print({k:v for k,v in d['roofline'].items() if not isinstance(v,str)})
"
