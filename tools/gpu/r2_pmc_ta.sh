#!/bin/bash
# which unit of the vector memory path is busy under the ring conv kernel (TA / TCP / TD / UTCL1)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r2g
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift; AERO_CONV_RING=2 timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc$n" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_conv.py" --layers d0 --iters 3 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.log" 2>&1; }
run 1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
run 2 TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run 3 TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum
run 4 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run 5 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r2g_pmc*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'conv' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(f.split('/')[1], k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -2 gpurun_out/${TAG}_pmc.log
