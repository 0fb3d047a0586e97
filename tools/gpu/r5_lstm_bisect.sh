#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for tag in "" lstm_pre1_imm0 lstm_pre0_imm1 lstm_pre0_imm0; do
  echo "== variant ${tag:-release (pre1 imm1)}"
  lib=""; [ -n "$tag" ] && lib=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_$tag.so
  AERO_HIP_LIB=$lib AERO_OLD_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_oldlstm.so timeout 200 python tools/dbg/lstm_check.py 2>&1 | grep "^H="
done
