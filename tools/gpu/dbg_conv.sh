#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-dbg}
mkdir -p gpurun_out
AERO_CONV_DEBUG=1 timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events 2> gpurun_out/${TAG}_convdbg.log | tail -1 | cut -c1-200
sort gpurun_out/${TAG}_convdbg.log | uniq -c | sort -rn | head -80 > gpurun_out/${TAG}_convdbg_uniq.log
for cfg in "AERO_LSTM_WIDE=0" "AERO_LSTM_WIDE=1"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/bench_lstm.py --iters 10 2>&1 | grep -v amdgpu.ids
done
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "lstm" -p no:cacheprovider 2>&1 | tail -2
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench.log
