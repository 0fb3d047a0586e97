"""One table per decoder 3x3 launch from rocprofv3 --pmc passes over tools/bench_conv.py (tools/gpu/r6_pmc_ring.sh): effective clock
(GRBM_GUI_ACTIVE / duration), MFMA-pipe busy share, LDS / wait shares, texture-path busy shares.  Writes profiles/pmc_clock.json (read by
bench.py for roofline.clock_ghz) when given --json.
usage: pmc_ring_table.py <counter_collection.csv ...> [--json out.json]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    out_json = None
    if '--json' in args:
        out_json = args[args.index('--json') + 1]
        args = [a for a in args if a not in ('--json', out_json)]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for path in args:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'aero_conv_ring_kernel' not in k:
                continue
            key = (k, r['Grid_Size'])
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r.get('Start_Timestamp') and r.get('End_Timestamp') and (r['Dispatch_Id'], path) not in seen:
                seen.add((r['Dispatch_Id'], path))
                durs[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    table = {}
    for key in sorted(acc, key=lambda k: -sum(durs[k]) if durs[k] else 0):
        m = {c: sum(v) / len(v) for c, v in acc[key].items()}
        d = sorted(durs[key])[len(durs[key]) // 2] if durs[key] else None            # median launch duration under the counters, ns
        gui = m.get('GRBM_GUI_ACTIVE')
        ghz = None
        if gui and d:
            ghz = gui / d
            if ghz > 4:                                       # summed over the 8 XCDs
                ghz /= 8
        wc = m.get('SQ_WAVE_CYCLES') or 0
        busy = m.get('SQ_BUSY_CYCLES') or 0
        row = {'grid': key[1], 'us': None if d is None else round(d / 1e3, 1), 'clock_ghz': None if ghz is None else round(ghz, 3)}
        if gui:
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles the MFMA pipe is busy, summed over SIMDs (guide: = 32 x N_mfma for 32x32x16); share of SIMD-cycles
            nsimd = 256 * 4
            g = gui / (8 if gui / max(d or 1, 1) > 4 else 1)
            if m.get('SQ_VALU_MFMA_BUSY_CYCLES'):
                row['mfma_busy_share'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * nsimd), 3)
            if m.get('SQ_INSTS_MFMA'):
                row['mfma_insts'] = m['SQ_INSTS_MFMA']
                row['mfma_issue_share_32cyc'] = round(m['SQ_INSTS_MFMA'] * 32 / (g * nsimd), 3)
        for c in ('SQ_WAIT_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY'):
            if c in m and wc:
                row[c.lower() + '_per_wave_cycle'] = round(m[c] / wc, 3)
        if 'SQ_LDS_BANK_CONFLICT' in m and m.get('SQ_LDS_IDX_ACTIVE'):
            row['lds_bank_conflict_share'] = round(m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'], 4)
        for c, name in (('TA_TA_BUSY_sum', 'ta_busy'), ('TCP_GATE_EN1_sum', 'tcp_gate'), ('TD_TD_BUSY_sum', 'td_busy'),
                        ('TCP_TCP_TA_DATA_STALL_CYCLES_sum', 'tcp_ta_data_stall'), ('TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'ta_addr_stalled_by_tc')):
            if c in m and gui:
                g = gui / (8 if gui / max(d or 1, 1) > 4 else 1)
                row[name + '_share'] = round(m[c] / (g * 256), 3)             # summed over the 256 CUs' units
        row['raw'] = {c: round(v) for c, v in m.items()}
        table[key[0] + ' grid ' + key[1]] = row
        print(key[0], 'grid', key[1])
        print('   ', {k: v for k, v in row.items() if k != 'raw'})
    if out_json:
        from bench import kernels_sha
        import datetime
        short = {}
        for k, row in table.items():
            name = k.split(' grid ')[0]
            if row.get('clock_ghz') and (name not in short or (row['us'] or 0) > (short[name]['us'] or 0)):
                short[name] = {'clock_ghz': row['clock_ghz'], 'us': row['us'], 'mfma_busy_share': row.get('mfma_busy_share')}
        short['_meta'] = {'kernels_sha': kernels_sha(), 'date': datetime.date.today().isoformat(), 'detail': table}
        json.dump(short, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main()
