#!/bin/bash
mkdir -p gpurun_out
for nb in 1 2 4; do
  AERO_STFT_DFT_BLOCKS=$nb python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > gpurun_out/r3g_bench_stft$nb.txt 2>&1
  python - <<PY
import json
d = json.loads(open('gpurun_out/r3g_bench_stft$nb.txt').read().strip().splitlines()[-1])
print('blocks $nb', d['ms_per_step'], {k[:40]: v for k, v in d['kernels_ms_per_step'].items() if 'stft' in k}, d['roofline_stft'])
PY
done
python tools/config5.py 2 4 2>&1 | grep step > gpurun_out/r3g_config5_b2.txt; cat gpurun_out/r3g_config5_b2.txt
python tools/config5.py 16 3 2>&1 | grep step > gpurun_out/r3g_config5_b16.txt; cat gpurun_out/r3g_config5_b16.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_disc.py -m gpu -q -k "stft or lstm_bwd or localstate_bwd or training_step or critic or grouped" 2>&1 | tail -4
