#!/bin/bash
# A/B of LSTM kernel variants + skinny conv (outputs under gpurun_out/)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ab}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_ab.log
: > $L
for cfg in "AERO_LSTM_RING=0" "AERO_LSTM_RING=1 AERO_LSTM_WIDE=0" "AERO_LSTM_RING=1 AERO_LSTM_WIDE=1"; do
  echo "== $cfg" >> $L
  env $cfg timeout 120 python tools/bench_lstm.py --iters 10 2>&1 | grep -v amdgpu.ids >> $L
done
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "lstm or conv" -p no:cacheprovider >> $L 2>&1
for cfg in "AERO_LSTM_RING=0 AERO_CONV_SKINNY=0" "AERO_LSTM_RING=0 AERO_CONV_SKINNY=1" "AERO_LSTM_RING=1 AERO_LSTM_WIDE=0" "AERO_LSTM_RING=1 AERO_LSTM_WIDE=1"; do
  echo "== bench $cfg" >> $L
  env $cfg timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> $L
done
cat $L
