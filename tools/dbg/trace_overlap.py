"""What ran beside what: a rocprofv3 --kernel-trace CSV of the serving loop (tools/dbg/sched_sweep.py with TRACE=1) reduced to
(i) the share of wall time with 0 / 1 / 2 / 3+ kernels in flight, (ii) per kernel family: launches, mean duration, the mean wait between the
end of the previous kernel of its queue and its own start, and the share of its duration during which a kernel of each OTHER family (from
another queue) was in flight.  The window is the steady state of the timed loop: from the end of the 24th-last batch (its iSTFT launch) to
the end of the 4th-last.
usage: trace_overlap.py <..._kernel_trace.csv> [label]"""
import collections
import csv
import sys

FAMILIES = (('ring256', ('aero_conv_ring_kernel<2, 4, 4', 'aero_conv_ring_kernel<2, 2, 4')), ('ring192', ('aero_conv_ring_kernel<2, 2, 3', 'aero_conv_ring_kernel<2, 4, 3')),
            ('ring-other', ('aero_conv_ring_kernel',)), ('lstm', ('aero_lstm',)), ('attn', ('aero_attn',)), ('glds8', ('aero_conv_glds8',)),
            ('glds4', ('aero_conv_glds_kernel', 'aero_conv_kernel')), ('pw', ('aero_pw_kernel',)), ('norm', ('aero_norm',)),
            ('rows', ('aero_enc0', 'aero_dconv_row')), ('fft', ('stft', 'aero_spec_normalize', 'aero_convtr_tail')))


def family(name):
    for f, pats in FAMILIES:
        if any(p in name for p in pats):
            return f
    return 'other'


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    rows, ends = [], []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), family(r['Kernel_Name']), r.get('Queue_Id', '')))
        if 'istft' in r['Kernel_Name']:
            ends.append(int(r['End_Timestamp']))
    rows.sort()
    ends.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    a, b = (ends[-24], ends[-4]) if len(ends) >= 24 else (t_lo, t_hi)
    nbatch = 20 if len(ends) >= 24 else 0
    sel = [r for r in rows if r[0] >= a and r[1] <= b]
    queues = sorted(set(r[3] for r in sel))
    print(f'== {label}: {len(sel)} kernels in the steady-state window of {(b - a) * 1e-6:.1f} ms' + (f' = {nbatch} batches, {(b - a) * 1e-6 / nbatch:.3f} ms per batch' if nbatch else '') + f', queues {queues}')
    ev = []
    for s, e, f, q in sel:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = collections.Counter()
    depth, last = 0, a
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        depth, last = depth + d, t
    tot = sum(hist.values())
    print('kernels in flight:  ' + '   '.join(f'{k}{"+" if k == 3 else ""}: {100 * hist[k] / tot:5.1f} %' for k in range(4)))
    # per-queue previous end
    prev_end = {}
    wait = collections.defaultdict(list)
    for s, e, f, q in sel:
        if q in prev_end:
            wait[f].append(s - prev_end[q])
        prev_end[q] = max(prev_end.get(q, 0), e)
    # overlap of family X with family Y on other queues: sweep over sorted intervals (n ~ 1e3-1e4: the quadratic pass is fine)
    dur = collections.defaultdict(list)
    beside = collections.defaultdict(lambda: collections.Counter())
    for i, (s, e, f, q) in enumerate(sel):
        dur[f].append(e - s)
        j = i - 1
        seen = collections.defaultdict(list)
        while j >= 0 and sel[j][0] > s - 3_000_000:
            s2, e2, f2, q2 = sel[j]
            if q2 != q and e2 > s:
                seen[f2].append((max(s, s2), min(e, e2)))
            j -= 1
        j = i + 1
        while j < len(sel) and sel[j][0] < e:
            s2, e2, f2, q2 = sel[j]
            if q2 != q:
                seen[f2].append((max(s, s2), min(e, e2)))
            j += 1
        for f2, iv in seen.items():
            iv.sort()
            cov, cur_s, cur_e = 0, None, None
            for x, y in iv:
                if cur_e is None or x > cur_e:
                    if cur_e is not None:
                        cov += cur_e - cur_s
                    cur_s, cur_e = x, y
                else:
                    cur_e = max(cur_e, y)
            if cur_e is not None:
                cov += cur_e - cur_s
            beside[f][f2] += cov
    nb = (b - a) * 1e-6
    print(f'{"family":10s} {"launches":>8s} {"mean us":>9s} {"sum ms":>8s} {"wait us":>8s}   share of its duration with another queue running ...')
    for f in [x[0] for x in FAMILIES] + ['other']:
        if not dur[f]:
            continue
        d = sum(dur[f])
        others = '  '.join(f'{f2} {100 * c / d:3.0f}%' for f2, c in beside[f].most_common(6))
        w = wait[f]
        print(f'{f:10s} {len(dur[f]):8d} {1e-3 * d / len(dur[f]):9.1f} {1e-6 * d:8.2f} {1e-3 * sum(w) / max(1, len(w)):8.1f}   {others}')
    print(f'(window {nb:.1f} ms; sum of kernel durations {1e-6 * sum(sum(v) for v in dur.values()):.1f} ms = {sum(sum(v) for v in dur.values()) / (b - a):.2f} x wall)')


if __name__ == '__main__':
    main()
