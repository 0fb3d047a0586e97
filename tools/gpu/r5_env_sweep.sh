#!/bin/bash
# pipelined step time under A/B switches that were neutral one batch at a time but change what can share a CU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r05_env_sweep.txt
: > $OUT
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))")
  echo "$tag: pipelined / one-at-a-time ms: $line" | tee -a $OUT
}
run default AERO_NOP=1
run ring_half2 AERO_RING_HALF=2
run lstm_narrow AERO_LSTM_WIDE=0
run ring_half2_lstm_narrow AERO_RING_HALF=2 AERO_LSTM_WIDE=0
run no_tile192 AERO_RING_TILE192=0
run depth2 AERO_PIPELINE=2
run depth4 AERO_PIPELINE=4
run depth6 AERO_PIPELINE=6
run default_again AERO_NOP=1
