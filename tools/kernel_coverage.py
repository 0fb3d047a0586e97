"""Which kernels of the shipped code object does anything LAUNCH?  (VERDICT r5 hygiene 8.)  Input: rocprofv3 --kernel-trace --stats
kernel_stats.csv files (the GPU test suite, the bench, the config-4 / config-5 runs); the code object's kernel list comes from its symbol
table.  Output: per kernel family the instantiations launched by (suite | bench configs) and those nothing launches.
    python tools/kernel_coverage.py --suite a_kernel_stats.csv --bench b_kernel_stats.csv [more ...]"""
import collections
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def code_object_kernels(lib):
    """demangled names of the kernels in the gfx950 code objects bundled in `lib` (their metadata notes, as tools/isa_lint.py reads them)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_lint
    notes = []
    isa_lint.disassemble(lib, notes)
    names = set()
    for txt in notes:
        names.update(re.findall(r'\.name:\s+(\S+)', txt))
    names = {n for n in names if n.startswith(('_Z', 'aero_'))}
    dm = isa_lint.demangle(sorted(names))
    return sorted(set(v.replace('void ', '').split('(')[0] for v in dm.values()))


def launched(path):
    s = set()
    for r in csv.DictReader(open(path)):
        n = r.get('Name') or r.get('Kernel_Name') or ''
        s.add(n.replace('void ', '').split('(')[0])
    return s


def main():
    groups = collections.OrderedDict()
    a = sys.argv[1:]
    i = 0
    while i < len(a):
        if a[i].startswith('--'):
            groups.setdefault(a[i][2:], set()).update(launched(a[i + 1]))
            i += 2
        else:
            i += 1
    kernels = code_object_kernels(os.path.join(ROOT, 'aero_amd', 'libaero_hip.so'))
    fam = collections.defaultdict(list)
    for k in kernels:
        fam[k.split('<')[0]].append(k)
    any_l = set().union(*groups.values()) if groups else set()
    print(f'{len(kernels)} kernels in the code object, {sum(1 for k in kernels if k in any_l)} launched by ' + ' | '.join(f'{g} ({len(v)})' for g, v in groups.items()))
    for f in sorted(fam, key=lambda f: -len(fam[f])):
        ks = fam[f]
        used = [k for k in ks if k in any_l]
        print(f'{f:40s} {len(ks):4d} instantiations, {len(used):3d} launched' + ''.join(f', {sum(1 for k in ks if k in v)} by {g}' for g, v in groups.items()))
        for k in ks:
            if k not in any_l:
                print(f'      never launched: {k}')


if __name__ == '__main__':
    main()
