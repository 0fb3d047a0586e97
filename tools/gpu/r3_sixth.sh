#!/bin/bash
mkdir -p gpurun_out
python - > gpurun_out/r3f_critic.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import disc_cases as dc
e = dc.case_critic_backward('cuda', T=8192)
for k, v in e.items():
    if v > 0.02 or not k.startswith('d.'):
        print(f'{k:50s} {v:.2e}')
PY
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -k "localstate or golden or batch64" 2>&1 | tail -3 > gpurun_out/r3f_pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_bench.txt 2>&1
grep -v Warn gpurun_out/r3f_critic.txt | tail -30; cat gpurun_out/r3f_pytest.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3f_bench.txt').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_conv_stack'])
for k, v in list(d['kernels_ms_per_step'].items())[:14]: print(f'{v:7.3f} {k[:90]}')
print(json.dumps(d['extra_configs'])[:1500])
PY
