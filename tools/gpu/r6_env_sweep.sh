#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r06_env_sweep_waits.txt
: > $out
for v in "X=0" "AERO_RING_HALF=2" "AERO_RING_192X128=0" "AERO_RING_256X128=1" "X=0" "AERO_LSTM_FRAME_MAJOR=0" "AERO_STFT_FUSED=0" "AERO_PITCHED_OUT=0" "AERO_PIPELINE_WAITS=0" "X=0"; do
  echo "$v: $(env $v timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep '^K=' | cut -c1-28 | tr '\n' ' ')" >> $out
done
cat $out
