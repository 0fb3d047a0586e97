#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_train" -o t -- python "$GRAFT_REPO_ROOT/tools/dbg/train_loop_nosync.py" 8 2>&1 | grep "ms per step"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_train" -name "*kernel_trace.csv" | head -1)
python "$GRAFT_REPO_ROOT/tools/step_timeline.py" "$f" > "$GRAFT_REPO_ROOT/gpurun_out/r06_train_timeline.txt" 2>&1
python - "$f" >> "$GRAFT_REPO_ROOT/gpurun_out/r06_train_timeline.txt" <<'PY'
import csv, sys, collections
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:60], r.get('Queue_Id', '')) for r in csv.DictReader(open(sys.argv[1])))
ends = [i for i, r in enumerate(rows) if 'aero_adam_kernel' in r[2]]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = step[0][0]
print('\n-- last step, per queue: busy ms, kernels; then the gaps > 15 us on the busiest queue')
by = collections.defaultdict(list)
for s, e, n, q in step: by[q].append((s, e, n))
for q, v in by.items():
    print(f'queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) * 1e-6:.2f} ms, first start {(v[0][0] - t0) * 1e-6:.2f} ms, last end {(max(e for _, e, _ in v) - t0) * 1e-6:.2f} ms')
main = max(by.values(), key=len)
gaps = [(main[i + 1][0] - main[i][1], main[i][2], main[i + 1][2], (main[i][1] - t0) * 1e-6) for i in range(len(main) - 1)]
print(f'main queue: sum of gaps {sum(g[0] for g in gaps) * 1e-6:.2f} ms over {len(gaps)} gaps; gaps > 15 us: {sum(1 for g in gaps if g[0] > 15000)} totalling {sum(g[0] for g in gaps if g[0] > 15000) * 1e-6:.2f} ms')
for g in sorted(gaps, reverse=True)[:25]:
    print(f'  {g[0] * 1e-3:7.1f} us at {g[3]:6.2f} ms  after {g[1]}  before {g[2]}')
PY
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r06_trace_train"
cat "$GRAFT_REPO_ROOT/gpurun_out/r06_train_timeline.txt" | cut -c1-200 | head -90
cd "$GRAFT_REPO_ROOT"
