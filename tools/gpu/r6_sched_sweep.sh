#!/bin/bash
# round 6: one scheduling variant per process, A/B against the plain schedule in the same process (tools/dbg/sched_sweep.py)
mkdir -p gpurun_out
out=gpurun_out/${OUT:-r06_sched_sweep.txt}
: > $out
for v in "lat-hi" "lat-hi-shared" "enc-hi" "dec-lo" "lat-hi+dec-lo" "enc-hi+dec-lo" "lat-mask64" "lat-mask64+dec-mask192" "dec-mask224" "lat-hi d4" "plain d4"; do
  echo "=== $v" >> $out
  ONLY="$v" timeout 300 python tools/dbg/sched_sweep.py ${K:-40} 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
