#!/bin/bash
# STFT / iSTFT A/B: parity on the GPU, then per-kernel time for ring sizes
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "stft" -p no:cacheprovider 2>&1 | tail -2
for v in ${AB_LIST:-default 32}; do
  if [ "$v" = default ]; then unset ${AB_VAR:-AERO_ISTFT_FPB}; else export ${AB_VAR:-AERO_ISTFT_FPB}=$v; fi
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_stft_$v.log
  python - "$v" <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/ab_stft_{sys.argv[1]}.log').read())
print('V', sys.argv[1], d['ms_per_step'], ' '.join(f'{k.replace("aero_","").replace("_kernel","")}={v:.4f}' for k, v in d['kernels_ms_per_step'].items() if 'stft' in k))
PY
done
