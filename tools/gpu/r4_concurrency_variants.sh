#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python tools/dbg/variants.py 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/r4c_variants.txt
cat gpurun_out/r4c_variants.txt
