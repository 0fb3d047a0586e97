#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
echo "projection floats behind the barrier (AERO_LSTM_PIN=0, rep $rep)"
AERO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_lstm_nopin.so timeout 200 python tools/bench_lstm.py --iters 20 2>&1 | grep "^H="
echo "projection pinned in front of the barrier (rep $rep)"
timeout 200 python tools/bench_lstm.py --iters 20 2>&1 | grep "^H="
done
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "lstm" 2>&1 | tail -2
