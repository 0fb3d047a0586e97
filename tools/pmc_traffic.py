"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into profiles/pmc_traffic.json:
{kernel name as rocprofv3 prints it: HBM-side bytes per launch}.  FETCH_SIZE is doubled as MI355X_MICROARCH.md
prescribes for gfx950 (128-byte requests tallied at 64 B); both counters are in KiB.
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith(('void aero_', 'aero_', '_Z')):
            continue
        name = k.replace('void ', '').split('(')[0]
        out[name] = {'fetch_bytes': 2.0 * fetch.get(k, 0.0) * 1024.0, 'write_bytes': write.get(k, 0.0) * 1024.0,
                     'bytes': (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0}
    import datetime
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernels_sha
    out['_meta'] = {'kernels_sha': kernels_sha(), 'date': datetime.date.today().isoformat(),
                    'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB -> bytes, average per launch'}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(f'{len(out)} kernels -> {sys.argv[3]}')


if __name__ == '__main__':
    main()
