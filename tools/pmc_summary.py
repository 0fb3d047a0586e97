"""Per-kernel summary of rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch and a few ratios.
python tools/pmc_summary.py gpurun_out/<tag>_pmc1/..csv [more csv ...]"""
import collections
import csv
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if not k.startswith('aero_'):
                continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
    rows = []
    for k, d in acc.items():
        m = {c: v / cnt[(k, c)] for c, v in d.items()}
        n = cnt[(k, 'SQ_WAVES')] or 1
        w = m.get('SQ_WAVES', 1.0) or 1.0
        rows.append((m.get('SQ_BUSY_CYCLES', 0) * n, k, n, w, m))
    print(f'{"kernel":44s} {"disp":>4s} {"waves":>8s} {"VALU/w":>7s} {"SALU/w":>7s} {"LDS/w":>6s} {"VMEM/w":>6s} {"MFMA/w":>6s} {"wait%":>6s} {"valu%":>6s} {"ldsw%":>6s}')
    for _, k, n, w, m in sorted(rows, reverse=True):
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1
        print(f'{k:44s} {n:4d} {w:8.0f} {m.get("SQ_INSTS_VALU", 0) / w:7.0f} {m.get("SQ_INSTS_SALU", 0) / w:7.0f} '
              f'{m.get("SQ_INSTS_LDS", 0) / w:6.0f} {(m.get("SQ_INSTS_VMEM_RD", 0) + m.get("SQ_INSTS_VMEM_WR", 0)) / w:6.0f} '
              f'{m.get("SQ_INSTS_MFMA", 0) / w:6.0f} {100 * m.get("SQ_WAIT_ANY", 0) / wc:6.1f} '
              f'{100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc:6.1f} {100 * m.get("SQ_WAIT_INST_LDS", 0) / wc:6.1f}')


if __name__ == '__main__':
    main()
