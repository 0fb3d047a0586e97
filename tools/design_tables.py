"""Print the measurement tables of DESIGN.md section 5 from a bench line (profiles/<tag>_bench.json) and the PMC traffic table
(profiles/<tag>_pmc_traffic.json).  usage: design_tables.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    d = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_bench.json')))
    pmc = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_traffic.json')))
    r = d['roofline']
    print(f"step {d['ms_per_step']} ms  RTF {d['value']}  dominant {r['kernel']} {r['achieved']} TF/s frac {r['frac']} avg {r['avg_launch_ms']} ms "
          f"traffic {r['traffic'] / 1e6 if r['traffic'] else None} MB")
    print('conv stack', d['roofline_conv_stack'])
    for k, v in d['roofline_stft'].items():
        print('stft', k, v['achieved'], 'GB/s', v['frac'], v['avg_launch_ms'], 'ms')
    print('cpu', d['cpu_baseline'])
    print('extra', json.dumps(d.get('extra_configs'), indent=1)[:1800])
    print()
    print('| kernel | ms / step | executed TFLOP/s (of 2500) | algorithmic TB/s (of 8.0) | PMC MB / launch |')
    print('|---|---|---|---|---|')
    for k, ms in d['kernels_ms_per_step'].items():
        tf, gb = d['kernels_achieved'][k]
        name = k.replace('void ', '').split('(')[0]
        ent = pmc.get(name)
        mb = f"{ent['bytes'] / 1e6:.0f}" if ent else ''
        tfs = f'{tf:.0f} ({100 * tf / 2500:.0f} %)' if tf else '—'
        print(f'| `{name}` | {ms:.3f} | {tfs} | {gb / 1000:.2f} ({100 * gb / 8000:.0f} %) | {mb} |')


if __name__ == '__main__':
    main()
