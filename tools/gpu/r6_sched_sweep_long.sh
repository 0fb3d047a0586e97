#!/bin/bash
# round 6: the scheduling variants in STEADY STATE (K = 200 batches per timing: the fill / drain of a 20-40 batch region is 0.2-0.4 ms per batch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r06_sched_sweep_long.txt
: > $out
for v in "lat-hi" "lat-fork" "lstm-hi" "enc-hi" "dec-lo" "lat-hi+dec-lo" "lat-hi d4"; do
  echo "=== $v" >> $out
  ONLY="$v" timeout 300 python tools/dbg/sched_sweep.py 200 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
