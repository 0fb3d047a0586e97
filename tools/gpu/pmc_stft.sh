#!/bin/bash
# PMC for the STFT / iSTFT kernels: LDS pipe activity, bank conflicts, waits (no trace domains combined with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/st_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/st_pmc$n.log" 2>&1; }
run 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run 2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for n in (1, 2):
    f = glob.glob(f'gpurun_out/st_pmc{n}/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv', n); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'stft' not in k: continue
        k = k.split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
