#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/dbg/pk_probe.py 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/r4d_pk_probe.txt
cat gpurun_out/r4d_pk_probe.txt
