#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${OUT:-r06_stagger_sweep2.txt}
: > $out
for d in ${DEPTHS:-3 4}; do for s in ${STAGES:-0 2.25 2.5 2.75 3 3.25 3.5 3.75}; do
  echo "=== depth $d stagger $s" >> $out
  DEPTH=$d STAGGER=$s timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep "^K=" >> $out
done; done
cat $out | cut -c1-170
