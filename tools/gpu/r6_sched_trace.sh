#!/bin/bash
# round 6: the control variants of the scheduling sweep + rocprofv3 kernel traces of the plain and the high-priority schedule
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r06_sched_sweep_b.txt
: > $out
for v in ${SWEEP}; do
  echo "=== $v" >> $out
  ONLY="$v" timeout 300 python tools/dbg/sched_sweep.py 40 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
cd /tmp
for v in plain lat-fork lat-hi lstm-hi; do
  tag=$(echo $v | tr -d '-')
  TRACE="$v" timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_$tag" -o t -- python "$GRAFT_REPO_ROOT/tools/dbg/sched_sweep.py" 30 2>&1 | grep "ms per batch"
  f=$(find "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_$tag" -name "*kernel_trace.csv" | head -1)
  python "$GRAFT_REPO_ROOT/tools/dbg/trace_overlap.py" "$f" "$v" > "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_overlap_$tag.txt"
  cat "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_overlap_$tag.txt"
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_$tag"
done
