#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r06f_pytest_gpu.log
cat gpurun_out/r06f_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06f_bench.json') if l.startswith('{')][0])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_one_at_a_time')})
print({k:v for k,v in d['roofline'].items() if not isinstance(v,str)})
print({k:v for k,v in d['config'].items() if k.startswith('config')})
"
timeout 300 python tools/config5.py 2 6 2>&1 | grep "^step" | tail -2
