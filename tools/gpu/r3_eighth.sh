#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "captured" 2>&1 | tail -25 > gpurun_out/r3h_pytest.txt
tail -25 gpurun_out/r3h_pytest.txt | cut -c1-600
python tools/profile_train.py 2 2>&1 | grep -v Warn | head -12
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > gpurun_out/r3h_bench.txt 2>&1
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3h_bench.txt').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d['extra_configs'][1])[:700])
PY
