#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu/pmc_stft.sh > gpurun_out/r4l_pmc_stft.txt 2>&1
cd "$GRAFT_REPO_ROOT"
for g in 1 2 3 4; do echo -n "AERO_STFT_DFT_BLOCKS=$g "; AERO_STFT_DFT_BLOCKS=$g timeout 100 python tools/launch_table.py 2>&1 | grep "stft_dft_kernel" | head -1; done > gpurun_out/r4l_dft_blocks.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l_table_default.txt
AERO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_nopk_all.so timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l_table_nopk_all.txt
tail -4 gpurun_out/r4l_pmc_stft.txt; cat gpurun_out/r4l_dft_blocks.txt
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'\s*(\d+)\s+([\d.]+) us.*?(aero_\S+|_Z\S+)', l)
        if m: d[int(m.group(1))]=(float(m.group(2)), m.group(3))
    return d
a,b=load('gpurun_out/r4l_table_default.txt'),load('gpurun_out/r4l_table_nopk_all.txt')
tot=0
for i in sorted(a):
    if i in b and abs(a[i][0]-b[i][0])>max(3,0.04*a[i][0]): print(i, a[i][1][:40], a[i][0], '->', b[i][0])
print('sum default', sum(v[0] for v in a.values()), 'nopk_all', sum(v[0] for v in b.values()))
PY
