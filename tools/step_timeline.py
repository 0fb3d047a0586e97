"""Timeline of the last training step in a rocprofv3 --kernel-trace CSV (tools/config5.py): wall time of the step, device busy time
(union of kernel intervals), idle gaps, time with more than one kernel in flight, and the kernels that account for the busy time.
usage: step_timeline.py <..._kernel_trace.csv> [marker kernel that ends a step = aero_adam_kernel]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'aero_adam_kernel'
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', ''))))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(ends) < 2:
        print('fewer than two steps in the trace')
        return
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    ev = []
    for s, e, _, _ in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = multi = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    print(f'step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.2f} ms, device busy {1e-6 * busy:.2f} ms, idle {1e-6 * (t1 - t0 - busy):.2f} ms, '
          f'two or more kernels in flight {1e-6 * multi:.2f} ms; queues {sorted(set(r[3] for r in step))}')
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _ in step:
        k = n.split('(')[0][:70]
        by[k][0] += 1
        by[k][1] += e - s
    tot = sum(v[1] for v in by.values())
    print(f'sum of kernel durations {1e-6 * tot:.2f} ms')
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{1e-6 * v[1]:7.3f} ms {v[0]:5d}  {k}')
    small = [e - s for s, e, _, _ in step if e - s < 8000]
    print(f'{len(small)} kernels shorter than 8 us: {1e-6 * sum(small):.2f} ms')


if __name__ == '__main__':
    main()
