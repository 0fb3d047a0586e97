#!/bin/bash
# L2 behaviour of the decoder 3x3 kernels: hit/miss, fabric fetch bytes, LDS/issue counters; ring vs k_conv.h tiles
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2e}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() { cfg=$1; n=$2; shift; shift; env $cfg timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_${cfg#*=}_pmc$n" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers ${LAYERS:-d0,d1} --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.log" 2>&1; }
for cfg in AERO_CONV_RING=0 AERO_CONV_RING=2; do
  run $cfg 1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run $cfg 2 FETCH_SIZE
  run $cfg 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES
  run $cfg 4 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r2e_*_pmc*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'conv' in k: acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(f.split('/')[1], k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -3 gpurun_out/${TAG}_pmc.log
