#!/bin/bash
# SQ counters of every kernel of one single-stream forward (separate --pmc passes; no trace domains)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference"
run() { n=$1; shift; AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/r06_fwdpmc$n" -o p -- python "$R/bench.py" $BARGS > "$R/gpurun_out/r06_fwdpmc$n.log" 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run 2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run 3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES
cd "$R"
python - <<'PY' > gpurun_out/r06_pmc_forward_table.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_fwdpmc*/**/*counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'aero' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if (r['Dispatch_Id'], f) not in seen and r['Counter_Name'] in ('SQ_WAVE_CYCLES',):
            seen.add((r['Dispatch_Id'], f)); dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = []
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(d.get('SQ_WAVES', d.get('SQ_WAVE_CYCLES', [1])))
    us = sum(dur[k]) / max(1, len(dur[k])) / 1e3
    rows.append((us * len(dur[k]) / 2, k, us, m))
print(f'{"kernel":46s} {"us":>7s} {"GHz":>5s} {"mfma%":>6s} {"valu%":>6s} {"lds%":>5s} {"conf%":>6s} {"waitI%":>6s} {"waitL%":>6s} {"VALU/w":>7s} {"LDS/w":>6s} {"VMEM/w":>6s} {"MFMA/w":>6s}')
for _, k, us, m in sorted(rows, reverse=True)[:40]:
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    w = m.get('SQ_WAVES', 0) or 1
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8
    ghz = gui / (us * 1e3) if us else 0
    mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) * 100 if gui else 0
    conf = 100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / (m.get('SQ_LDS_IDX_ACTIVE', 0) or 1)
    print(f'{k[:46]:46s} {us:7.1f} {ghz:5.2f} {mf:6.1f} {100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc:6.1f} {100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc:5.1f} {conf:6.1f} '
          f'{100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:6.1f} {100 * m.get("SQ_WAIT_INST_LDS", 0) / wc:6.1f} {m.get("SQ_INSTS_VALU", 0) / w:7.0f} {m.get("SQ_INSTS_LDS", 0) / w:6.0f} '
          f'{(m.get("SQ_INSTS_VMEM_RD", 0) + m.get("SQ_INSTS_VMEM_WR", 0)) / w:6.0f} {m.get("SQ_INSTS_MFMA", 0) / w:6.0f}')
PY
cat gpurun_out/r06_pmc_forward_table.txt
rm -rf gpurun_out/r06_fwdpmc1 gpurun_out/r06_fwdpmc2 gpurun_out/r06_fwdpmc3
