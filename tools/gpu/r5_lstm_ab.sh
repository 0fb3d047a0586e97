#!/bin/bash
# LSTM ring kernel A/B: round-4 step loop (tools/dbg/libaero_hip_oldlstm.so) vs pre-scaled weights + immediate slot addressing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r05_lstm_ab.txt
: > $OUT
for rep in 1 2; do
echo "old kernel (rep $rep)" >> $OUT
AERO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_oldlstm.so timeout 200 python tools/bench_lstm.py --iters 20 2>&1 | grep "^H=" >> $OUT
echo "new kernel (rep $rep)" >> $OUT
timeout 200 python tools/bench_lstm.py --iters 20 2>&1 | grep "^H=" >> $OUT
done
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "blstm or lstm or module_vectors or golden or train" 2>&1 | tail -4 >> $OUT
for lib in $GRAFT_REPO_ROOT/tools/dbg/libaero_hip_oldlstm.so ""; do
  AERO_HIP_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench lib=${lib##*/}', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))" >> $OUT
done
cat $OUT
