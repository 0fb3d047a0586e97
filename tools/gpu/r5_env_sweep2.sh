#!/bin/bash
# pipelined step time under tile-selection switches: which choices made one batch at a time still hold with three batches in flight?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r05_env_sweep2.txt
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
run conv_bm256_off AERO_CONV_BM256=0
run conv_bm256_only256 AERO_CONV_BM256=1
run ring_kmin256_1024 AERO_RING_KMIN256=1024
run conv_kmin192_768 AERO_CONV_KMIN192=768
run fuse_stats_always AERO_FUSE_STATS=1
run fuse_stats_never AERO_FUSE_STATS=0
run convtr_stack_off AERO_CONVTR_STACK=0
run default_again AERO_NOP=1
run ring_half2 AERO_RING_HALF=2
run tap_split_off AERO_TAP_SPLIT=0
run default_3 AERO_NOP=1
