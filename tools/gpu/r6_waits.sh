#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${OUT:-r06_waits_sweep.txt}
: > $out
for d in ${DEPTHS:-3}; do for w in ${WAITSET:-"" "0:3" "4:8" "0:3,4:8" "0:2.5,4:8" "0:3,4:7" "4:7" "0:3,5:8" "0:2,4:8" "3:8" "0:3,3:8"}; do
  echo "=== depth $d waits $w" >> $out
  DEPTH=$d WAITS=$w timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep "^K=" | cut -c1-120 >> $out
done; done
cat $out
