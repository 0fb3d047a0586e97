#!/bin/bash
# STFT / iSTFT: parity on the GPU, then their per-kernel time from a short bench run
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "stft" -p no:cacheprovider 2>&1 | tail -2
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_stft.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/ab_stft.log').read())
print(d['ms_per_step'], ' '.join(f'{k.replace("aero_","").replace("_kernel","")}={v:.4f}' for k, v in d['kernels_ms_per_step'].items() if 'stft' in k or 'spec_norm' in k))
PY
