#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "AERO_X=0" "AERO_LSTM_RING=0" "AERO_CONV_SKINNY=0" "AERO_CONV_BM256=0" "AERO_CONV_GLDS=0"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/bisect_batch.py 2>&1 | grep -v amdgpu.ids | tail -2
done
echo "== no-ftb0"; timeout 200 python tools/bisect_batch.py --no-ftb0 2>&1 | grep -v amdgpu.ids | tail -2
