"""Fill the @@PLACEHOLDER@@ tokens of DESIGN.md section 4.8 / 4.9 / 5 (and README.md) from the evidence files of a visit:
profiles/<tag>_bench.json, <tag>_config5_b2.txt, <tag>_config5_b16.txt, <tag>_config5_gan_b2.txt, <tag>_train_profile_b2.txt,
<tag>_pytest_gpu.log.  usage: fill_design.py r03 [--check]   (--check: only list what would be filled)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_steps(path, n=2):
    rows = [l for l in open(path) if l.startswith('step ')]
    return rows[-n:]


def mean_field(rows, pat):
    v = [float(re.search(pat, r).group(1)) for r in rows]
    return sum(v) / len(v)


FWD, BWD, CRIT = r'forward ([0-9.]+) ms', r'loss\+backward ([0-9.]+) ms', r'critic step ([0-9.]+) ms'


def main():
    tag = sys.argv[1]
    P = lambda name: os.path.join(ROOT, 'profiles', f'{tag}_{name}')   # noqa: E731
    d = json.load(open(P('bench.json')))
    r, cs, st = d['roofline'], d['roofline_conv_stack'], d['roofline_stft']
    stft = next(v for k, v in st.items() if 'stft_dft' in k or ('stft' in k and 'istft' not in k))
    istft = next(v for k, v in st.items() if 'istft' in k)
    ex = d.get('extra_configs') or []
    c4 = next((e for e in ex if 'config 4' in json.dumps(e)), {})
    c5 = next((e for e in ex if 'ms_per_step_hip_graph' in e), {})
    c5g = next((e for e in ex if 'adversarial training steps' in e.get('metric', '')), {})
    v = {
        'MS': f"{d['ms_per_step']:.2f}", 'RTF': f"{d['value']:,.0f}".replace(',', ' '),
        'CPU': f"{d['cpu_baseline']['value']:.2f}", 'CPU1': f"{d['cpu_baseline'].get('value_1_thread', float('nan')):.2f}",
        'DOMMS': f"{r['avg_launch_ms']:.3f}", 'DOMTF': f"{r['achieved']:.0f}", 'DOMFRAC': f"{100 * r['frac']:.1f}",
        'TRAFFIC': f"{r['traffic'] / 1e9:.2f}" if r.get('traffic') else 'n/a',
        'STACKTF': f"{cs['achieved']:.0f}", 'STACKFRAC': f"{100 * cs['frac']:.1f}", 'STACKMS': f"{cs['ms_per_step']:.2f}",
        'STFTUS': f"{1e3 * stft['avg_launch_ms']:.0f}", 'STFTFRAC': f"{100 * stft['frac']:.0f}",
        'ISTFTUS': f"{1e3 * istft['avg_launch_ms']:.0f}", 'ISTFTFRAC': f"{100 * istft['frac']:.0f}",
        'C4MS': f"{c4.get('ms_per_step', float('nan')):.2f}", 'C4RTF': f"{c4.get('value', float('nan')):,.0f}".replace(',', ' '),
        'C5MS': f"{c5.get('ms_per_step', float('nan')):.1f}", 'C5G': str(c5.get('ms_per_step_hip_graph')),
        'C5GAN': f"{c5g.get('ms_per_step', float('nan')):.1f}",
    }
    if os.path.exists(P('config5_b2.txt')):
        rows = last_steps(P('config5_b2.txt'))
        v['T2FWD'] = f"{mean_field(rows, FWD):.1f}"
        v['T2BWD'] = f"{mean_field(rows, BWD):.1f}"
    if os.path.exists(P('config5_b16.txt')):
        rows = last_steps(P('config5_b16.txt'))
        f, b = mean_field(rows, FWD), mean_field(rows, BWD)
        v['T16FWD'], v['T16BWD'], v['T16'] = f'{f:.1f}', f'{b:.1f}', f'{f + b + 0.5:.0f}'
    if os.path.exists(P('config5_gan_b2.txt')):
        rows = last_steps(P('config5_gan_b2.txt'))
        v['GANGEN'] = f"{mean_field(rows, BWD):.1f}"
        v['GANCRIT'] = f"{mean_field(rows, CRIT):.1f}"
    if os.path.exists(P('train_profile_b2.txt')):
        m = re.search(r'([0-9.]+) ms\s+\d+ calls\s+aero_conv_wgrad', open(P('train_profile_b2.txt')).read())
        if m:
            v['WGRADMS'] = f'{float(m.group(1)):.1f}'
    if os.path.exists(P('pytest_gpu.log')):
        m = re.findall(r'(\d+) passed', open(P('pytest_gpu.log')).read())
        if m:
            v['NGPU'] = m[-1]
    for k in sorted(v):
        print(f'{k:10s} {v[k]}')
    if '--check' in sys.argv:
        return
    for name in ('DESIGN.md', 'README.md'):
        path = os.path.join(ROOT, name)
        s = open(path).read()
        for k, val in v.items():
            s = s.replace(f'@@{k}@@', val)
        left = sorted(set(re.findall(r'@@[A-Z0-9]+@@', s)))
        if left:
            print(name, 'still has', left)
        open(path, 'w').write(s)


if __name__ == '__main__':
    main()
