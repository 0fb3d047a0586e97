#!/bin/bash
# SQ / LDS / issue counters of one micro-benchmarked kernel:  r2_pmc_op.sh TAG tools/bench_enc0.py enc0
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2op}; SCRIPT=${2:-tools/bench_enc0.py}; PAT=${3:-enc0}
mkdir -p gpurun_out
export TMPDIR=/tmp
python "$SCRIPT" > gpurun_out/${TAG}_time.log 2>&1
cd /tmp
run() { n=$1; shift; ITERS=2 timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p -- python "$GRAFT_REPO_ROOT/$SCRIPT" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.log" 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES
run 2 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU
run 3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS
run 4 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM
cd "$GRAFT_REPO_ROOT"
cat gpurun_out/${TAG}_time.log | tail -3
python - "$TAG" "$PAT" <<'PY'
import csv, glob, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f'gpurun_out/{tag}_pmc*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if pat in k: acc[(k, r['Grid_Size'], r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'))][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(f.split('/')[1], k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -3 gpurun_out/${TAG}_pmc.log
