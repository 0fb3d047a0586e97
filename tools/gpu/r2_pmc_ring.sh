#!/bin/bash
# PMC passes over the decoder 3x3 shapes with the ring kernel (separate passes; no trace domains with --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2c}
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/dbg/batch_invariance.py wide 24000 32 > gpurun_out/dbg_binv_wide32.log 2>&1
cd /tmp
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p$n -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0,d2,d3 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n.log" 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run 2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run 3 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
# where do the __amd_rocclr_copyBuffer dispatches of a forward come from?  (HIP API trace + kernel trace, no PMC)
timeout 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/${TAG}_pmc1/*/*counter_collection.csv gpurun_out/${TAG}_pmc2/*/*counter_collection.csv gpurun_out/${TAG}_pmc3/*/*counter_collection.csv 2>&1 | tail -8
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r2c_pmc*/*/*counter_collection.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'conv' in k: acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
ls gpurun_out/${TAG}_trace/*/ | head; f=$(ls gpurun_out/${TAG}_trace/*/*hip_api_trace.csv | head -1); python - "$f" <<'PY'
import csv, sys, collections
c = collections.Counter(r['Function'] for r in csv.DictReader(open(sys.argv[1])))
print(c.most_common(12))
PY
tail -12 gpurun_out/dbg_binv_wide32.log
