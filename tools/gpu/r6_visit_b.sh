#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_concurrency.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r06b_pytest.log
cat gpurun_out/r06b_pytest.log
for w in none auto none auto; do
  if [ $w = auto ]; then unset WAITS; else export WAITS=$w; fi
  echo "waits=$w: $(timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep '^K=' | cut -c1-60 | tr '\n' ' ')"
done | tee gpurun_out/r06b_waits_ab.txt
unset WAITS
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06b_bench.json') if l.startswith('{')][0])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_one_at_a_time')})
print({k:v for k,v in d['roofline'].items() if not isinstance(v,str)})
print({k:v for k,v in d['config'].items() if k.startswith('config')})
print(d['cpu_baseline'])
"
