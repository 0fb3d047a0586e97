"""ISA lint of the gfx950 code object inside aero_amd/libaero_hip.so: a cheap guard for the hazard class that
produced the two hardware-only wrong results of round 1 -- kernels that synchronise with RAW `s_barrier`s and stage
operands with direct global->LDS copies, where (i) hipcc may schedule a register-only MFMA into the window between the LDS wait and
the barrier that closes a phase, and (ii) a ring slot may be refilled by a `global_load_lds` that is not separated by a barrier from the
last LDS reads of the previous phase.

Per kernel it reports: barriers, direct-to-LDS copies, MFMAs, and
  W1  MFMAs that sit between an `s_waitcnt ... lgkmcnt(0)` and the `s_barrier` that follows it with no LDS read in between
      (the compiler moved matrix work into the wait -> barrier window);
  W2  direct-to-LDS copies issued after LDS reads with NO `s_barrier` between the last `ds_read` and the copy (same basic block).
Neither is an error by itself (W1 is harmless when the MFMA's operands were read before the wait; W2 when the slot is not the one being
read): the report is for review whenever a kernel with rings of LDS slots changes, and `--strict K1,K2` fails if a named kernel's counts grow
over the committed baseline (profiles/isa_lint_baseline.json).  It also counts the packed-fp32 VALU instructions of the whole code
object and exits with code 3 if there is one (the build fails on that: DESIGN.md 5b).
W4: kernels with a private segment outside SCRATCH_ALLOWED (exit code 5; see the comment there).
usage: isa_lint.py [lib.so] [--write-baseline] [--strict] [--allow-packed]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(lib, notes=None):
    """the gfx950 code objects of EVERY fat-binary bundle in `lib` (the library is linked from seven separately compiled parts, each with
    its own bundle in .hip_fatbin: unbundling the section as a whole yields only the first one), disassembled and concatenated;
    notes: a list that receives each code object's `llvm-readelf --notes` text (kernel metadata)"""
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run([f'{LLVM}/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        assert starts, f'{lib}: no offload bundle in .hip_fatbin'
        for i, a in enumerate(starts):
            part, co = os.path.join(td, f'bundle{i}.bin'), os.path.join(td, f'gfx950_{i}.co')
            open(part, 'wb').write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([f'{LLVM}/clang-offload-bundler', '--type=o', f'--input={part}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                            f'--output={co}', '--unbundle'], check=True)
            out.append(subprocess.run([f'{LLVM}/llvm-objdump', '-d', co], check=True, capture_output=True, text=True).stdout)
            if notes is not None:
                notes.append(subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], check=True, capture_output=True, text=True).stdout)
    return '\n'.join(out)


# Kernels with a PRIVATE SEGMENT (spilled registers, indexed local arrays).  Measured in round 5 (profiles/r05_noscratch.txt): a launch of
# such a kernel is ~40 us slower whenever it needs more scratch than the stream's queue holds at that moment -- the first one of every
# forward, and every later one that needs more than its predecessors (an 8-wave conv tile with 3 spilled registers 160 -> 115 us, the LSTM
# with 8 spilled dwords 193 -> 148 us; the same kernel 40 launches later, with the scratch already there: no difference).  No kernel of the
# measured paths may have one; these are the instantiations outside them that still do (rule W4, exit code 5 for anything else):
SCRATCH_ALLOWED = (r'^aero_pw_kernel<2, 3, [01], false, false>$', r'^aero_lstm_kernel<8, 3, 3, 6>$', r'^aero_conv_glds_kernel<4, 2, 32, true>$',
                   r'^aero_conv_glds8_kernel<4, 32, true>$', r'^aero_conv_skinny_kernel<[1248]>$')


def private_segments(notes):
    """{mangled kernel name: private_segment_fixed_size} for every kernel of the code objects' metadata that has one"""
    out = {}
    for txt in notes:
        for blk in re.split(r'\n\s*- \.agpr_count', txt)[1:]:
            nm, ps = re.search(r'\.name:\s+(\S+)', blk), re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
            if nm and ps and int(ps.group(1)):
                out[nm.group(1)] = int(ps.group(1))
    return out


def lint(text):
    out = {}
    name, ins = None, []

    def flush():
        if name is None or not ins:
            return
        nbar = sum(i.startswith('s_barrier') for i in ins)
        nglds = sum(('global_load_lds' in i) or (i.startswith('buffer_load') and ' lds' in i) for i in ins)
        nmfma = sum(i.startswith('v_mfma') for i in ins)
        w1 = w2 = 0
        in_window, mf_in_window = False, 0
        reads_since_barrier = 0
        for i in ins:
            if i.startswith('s_waitcnt') and 'lgkmcnt(0)' in i:
                in_window, mf_in_window = True, 0
            elif i.startswith('ds_read') or i.startswith('ds_load'):
                in_window = False
                reads_since_barrier += 1
            elif i.startswith('v_mfma') and in_window:
                mf_in_window += 1
            elif i.startswith('s_barrier'):
                if in_window:
                    w1 += mf_in_window
                in_window = False
                reads_since_barrier = 0
            elif ('global_load_lds' in i) or (i.startswith('buffer_load') and ' lds' in i):
                if reads_since_barrier:
                    w2 += 1
            elif i.startswith(('s_cbranch', 's_branch', 's_endpgm')):
                in_window = False
                if not i.startswith('s_cbranch'):               # no fall-through: what follows in the text is another path (e.g. the prologue of a
                    reads_since_barrier = 0                     # second instantiation of a pipelined loop), not the continuation of these reads
        if nbar and (nglds or nmfma):
            out[name] = {'barriers': nbar, 'lds_copies': nglds, 'mfma': nmfma, 'W1_mfma_in_wait_barrier_window': w1,
                         'W2_lds_copy_after_reads_without_barrier': w2}
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            flush()
            name, ins = m.group(1), []
            continue
        s = line.strip()
        if s and not s.startswith(('//', 'Disassembly', 'gfx')) and name is not None:
            ins.append(s.split('//')[0].strip())
    flush()
    return out


def _regs(tok):
    """VGPR indices named by an operand token: v12, v[12:15]"""
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'^v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def mfma_branch_hazards(text, min_wait=2):
    """W3 (round 5): a VALU instruction at a BRANCH TARGET that reads the destination of a matrix instruction issued right in front of the
    branch, with fewer than `min_wait` wait states in between (2: the rule looks for paths
    the recogniser did not pad at all, not for the exact count a given matrix shape needs).  hipcc's hazard recogniser pads the fall-through path of such a diamond and
    has been seen to leave the taken path bare (`v_mfma ... ; s_cbranch_execnz L ; ... L: v_mov_b64 <mfma dst>`): the first unrolled LSTM
    step of round 5 returned run-to-run different results on the MI355X because of exactly that.  Returns
    {kernel: count}; any hit FAILS the lint."""
    out = {}
    for blk in re.split(r'\n(?=[0-9a-f]{16} <)', text):
        m = re.match(r'[0-9a-f]{16} <(\S+)>:', blk)
        if not m:
            continue
        ins = []
        for line in blk.splitlines()[1:]:
            mm = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', line)
            if mm:
                ins.append((int(mm.group(3), 16), mm.group(1), [t.strip() for t in mm.group(2).split(',') if t.strip()]))
        at = {a: i for i, (a, _, _) in enumerate(ins)}
        hits = 0
        for i, (a, op, args) in enumerate(ins):
            if not (op.startswith('s_cbranch') or op == 's_branch') or not args:
                continue
            try:
                off = int(args[0].split()[0])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            j = at.get(a + 4 + off * 4)
            if j is None:
                continue
            # matrix results still in flight at the branch: MFMAs among the last three instructions in front of it
            dst, waited = set(), 0
            for k in range(i - 1, max(-1, i - 4), -1):
                o, ar = ins[k][1], ins[k][2]
                if o.startswith('v_mfma') and ar:
                    dst |= _regs(ar[0])
                    break
                if o == 's_nop' and ar:
                    waited += int(ar[0]) + 1
                elif not o.startswith('s_'):
                    waited += 1
            if not dst:
                continue
            w = waited
            for k in range(j, min(len(ins), j + 6)):
                o, ar = ins[k][1], ins[k][2]
                if o == 's_nop' and ar:
                    w += int(ar[0]) + 1
                    continue
                if o.startswith('v_') and not o.startswith('v_mfma'):
                    if any(_regs(t) & dst for t in ar[1:]):
                        if w < min_wait:
                            hits += 1
                        break
                    if ar:
                        dst -= _regs(ar[0])                       # overwritten by this VALU: later readers see ITS value
                        if not dst:
                            break
                if o.startswith(('s_cbranch', 's_branch', 's_barrier', 's_waitcnt')):
                    break
                w += 1
        if hits:
            out[m.group(1)] = hits
    return out


DEMANGLER_OK = True


def demangle(names):
    """llvm-cxxfilt of the ROCm toolchain first, then the system's c++filt.  With neither, the names stay mangled and DEMANGLER_OK goes
    False: the checks that match kernels by their demangled names (scratch allow-list, baseline comparison) are then SKIPPED with a
    warning instead of reporting every kernel as a violation."""
    global DEMANGLER_OK
    names = list(names)
    if not names:
        return {}
    for tool in (os.path.join(LLVM, 'llvm-cxxfilt'), 'llvm-cxxfilt', 'c++filt'):
        try:
            p = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True, check=True)
            out = p.stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            continue
    DEMANGLER_OK = False
    return {n: n for n in names}


PACKED_FP32 = re.compile(r'^\s*(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32)\b', re.M)


def main():
    lib = os.path.join(ROOT, 'aero_amd', 'libaero_hip.so')
    for a in sys.argv[1:]:
        if a.endswith('.so'):
            lib = a
    notes = []
    text = disassemble(lib, notes)
    priv = private_segments(notes)
    pdm = demangle(list(priv))
    priv = {pdm[k].split('(')[0].replace('void ', ''): v for k, v in priv.items()}
    priv_bad = sorted(k for k in priv if not any(re.match(a, k) for a in SCRATCH_ALLOWED)) if DEMANGLER_OK else []
    # packed-fp32 VALU instructions anywhere in the code object: with them the FFT-family kernels return wrong values next to another
    # stream's MFMA waves; the library is built without them, and this count FAILS the build (exit code 3) if it is not 0
    npk = len(PACKED_FP32.findall(text))
    w3 = mfma_branch_hazards(text)
    rep = lint(text)
    dm = demangle(list(rep))
    rep = {dm[k].split('(')[0].replace('void ', ''): v for k, v in rep.items()}
    base_path = os.path.join(ROOT, 'profiles', 'isa_lint_baseline.json')
    if '--write-baseline' in sys.argv:
        json.dump(rep, open(base_path, 'w'), indent=1, sort_keys=True)
    bad = []
    if not DEMANGLER_OK:
        print('WARNING: no demangler (llvm-cxxfilt / c++filt): scratch allow-list and baseline comparison skipped')
    if os.path.exists(base_path) and DEMANGLER_OK:
        base = json.load(open(base_path))
        for k, v in rep.items():
            b = base.get(k)
            if b and (v['W1_mfma_in_wait_barrier_window'] > b['W1_mfma_in_wait_barrier_window'] or
                      v['W2_lds_copy_after_reads_without_barrier'] > b['W2_lds_copy_after_reads_without_barrier']):
                bad.append((k, b, v))
    print(f'{len(rep)} kernels with barriers + (MFMA | direct-to-LDS copies)')
    print(f'{"kernel":70s} {"bar":>4s} {"glds":>5s} {"mfma":>5s} {"W1":>4s} {"W2":>4s}')
    for k, v in sorted(rep.items(), key=lambda kv: -(kv[1]['W1_mfma_in_wait_barrier_window'] + kv[1]['W2_lds_copy_after_reads_without_barrier'])):
        print(f'{k[:70]:70s} {v["barriers"]:4d} {v["lds_copies"]:5d} {v["mfma"]:5d} {v["W1_mfma_in_wait_barrier_window"]:4d} {v["W2_lds_copy_after_reads_without_barrier"]:4d}')
    print(f'PACKED_FP32 {npk} v_pk_{{fma,mul,add}}_f32 instructions in the code object (must be 0)')
    w3d = demangle(list(w3))
    print(f'MFMA_BRANCH_HAZARD {sum(w3.values())} VALU reads of a matrix result at a branch target without wait states (must be 0)'
          + (': ' + ', '.join(f'{w3d[k].split("(")[0]} x{v}' for k, v in w3.items()) if w3 else ''))
    print(f'PRIVATE_SEGMENT {len(priv)} kernels use scratch memory, {len(priv_bad)} of them outside the allowed list (must be 0)'
          + (': ' + ', '.join(f'{k} ({priv[k]} B)' for k in priv_bad) if priv_bad else ''))
    if bad:
        print('GREW over the baseline:', [b[0] for b in bad])
    if npk and '--allow-packed' not in sys.argv:
        sys.exit(3)
    if w3:
        sys.exit(4)
    if priv_bad:
        sys.exit(5)
    if bad and '--strict' in sys.argv:
        sys.exit(1)


if __name__ == '__main__':
    main()
