#!/bin/bash
# stagger sweep of the two-stream forward (+ 3 / 4 sub-batches), pw chunk order
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events"
run() { echo -n "$1: "; env $1 timeout 120 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
{
run AERO_STAGGER=0
run AERO_STREAMS=1
for s in 1 2 3 4 5 6 7; do run AERO_STAGGER=$s; done
run AERO_STAGGER=0
for s in 2 3 4 5; do run "AERO_STREAMS=3 AERO_STAGGER=$s"; done
for s in 0 2 3; do run "AERO_STREAMS=4 AERO_STAGGER=$s"; done
run AERO_STAGGER=0
} > gpurun_out/r4g_stagger.txt 2>&1
timeout 200 python tools/launch_table.py 2>&1 | grep "aero_pw\|sum of" > gpurun_out/r4g_pw.txt
cat gpurun_out/r4g_stagger.txt gpurun_out/r4g_pw.txt
