#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -12 > gpurun_out/r4m_pytest.txt
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4m_table_default.txt
AERO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/libaero_hip_nopk_all.so timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4m_table_nopk_all.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
for i in 1 2 3; do timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done > gpurun_out/r4m_bench.txt
cat gpurun_out/r4m_pytest.txt
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'\s*(\d+)\s+([\d.]+) us.*?(aero_\S+|_Z\S+)', l)
        if m: d[int(m.group(1))]=(float(m.group(2)), m.group(3))
    return d
a,b=load('gpurun_out/r4m_table_default.txt'),load('gpurun_out/r4m_table_nopk_all.txt')
for i in sorted(a):
    if i in b and abs(a[i][0]-b[i][0])>max(3,0.04*a[i][0]): print(i, a[i][1][:40], a[i][0], '->', b[i][0])
print('sum default', round(sum(v[0] for v in a.values())), 'nopk_all', round(sum(v[0] for v in b.values())))
PY
grep "aero_pw" gpurun_out/r4m_table_default.txt | cut -c1-140; cat gpurun_out/r4m_bench.txt
