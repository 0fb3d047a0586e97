"""Round 4: instruction-class probe for the cross-stream wrong-result finding (tools/dbg/pk_probe.hip).  Each probe kernel is a
register-only dependent chain of one VALU instruction class; it runs 100 times next to (a) the 192-row ring conv tile, (b) a rocBLAS GEMM,
(c) nothing, and every result is compared bit for bit with the solo launch.  Reports mismatching rounds and which LANES were wrong."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
SO = os.path.join(ROOT, 'tools', 'dbg', 'libpk_probe.so')
NAMES = {12: 'v_pk_fma_f32 ; s_mov exec (lanes 1-63 off)', 13: 'v_fma_f32 x2 ; s_mov exec (control)', 8: 'v_pk_mul_f32 V, V, SGPR pair', 9: 'v_pk_mul_f32 V, V, 0.5', 10: 'v_pk_add_f32 V, V, 0 neg', 11: 'v_pk_fma_f32 V, V, 0.5, V', 6: 'C++ butterflies through LDS (packed ops)', 7: 'C++ butterflies through LDS (built without packed-fp32)', 0: 'v_fma_f32 (control)', 1: 'v_pk_fma_f32', 2: 'v_pk_fma_f32 op_sel/neg (cmul form)', 3: 'v_pk_mul_f32 + v_pk_add_f32', 4: 'v_pk_fma_f16',
         5: 'v_pk_fma_f32 around v_exp_f32'}


def main():
    if '--build' in sys.argv:
        src = os.path.join(ROOT, 'tools', 'dbg', 'pk_probe.hip')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO, src], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops', '-DNO_PK_ASM',
                        '-o', SO.replace('.so', '_nopk.so'), src], check=True)
        return
    import torch
    import concurrency_cases as cc
    from aero_amd import _lib
    lib = C.CDLL(SO)
    lib.pk_probe.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib_nopk = C.CDLL(SO.replace('.so', '_nopk.so'))
    lib_nopk.pk_probe.argtypes = lib.pk_probe.argtypes
    dist = cc.RingDisturber(_lib.load(), 'cuda')
    mm_a = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
    nblocks, iters = 4096, 16000
    out = torch.zeros(nblocks * 256 * 2, device='cuda')

    def d_mm(n):
        for _ in range(n):
            mm_a @ mm_a
    plain = {}
    for mode in (1, 12, 0, 13):
        L_ = lib_nopk if mode == 7 else lib

        def victim():
            o = torch.empty_like(out)
            L_.pk_probe(6 if mode == 7 else mode, o.data_ptr(), nblocks, iters, torch.cuda.current_stream().cuda_stream)
            return o
        ref = victim().clone()
        torch.cuda.synchronize()
        plain[mode] = ref
        if mode == 12:
            print('   solo: exec-write variant == plain v_pk_fma chain:', torch.equal(ref, plain[1]))
        if mode == 13:
            print('   solo: exec-write control == plain v_fma chain:', torch.equal(ref, plain[0]))
        line = f'{NAMES[mode]:40s}'
        for dname, dfn, nd in (('ring192', dist.launch, 4), ('rocblas', d_mm, 3), ('idle', lambda n: None, 0)):
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            bad, lanes = 0, torch.zeros(64, dtype=torch.int64, device='cuda')
            halves = [0, 0]
            for it in range(100):
                cur = torch.cuda.current_stream()
                sa.wait_stream(cur)
                sb.wait_stream(cur)
                with torch.cuda.stream(sa):
                    dfn(nd)
                with torch.cuda.stream(sb):
                    o = victim()
                torch.cuda.synchronize()
                ne = (o != ref).view(-1, 2)
                if bool(ne.any()):
                    bad += 1
                    idx = ne.any(1).nonzero().flatten()
                    lanes += torch.bincount(idx % 64, minlength=64)
                    halves[0] += int(ne[:, 0].sum())
                    halves[1] += int(ne[:, 1].sum())
            line += f' | {dname}: {bad:3d}/100'
            if bad:
                nzl = lanes.nonzero().flatten().tolist()
                line += f' lanes {nzl[0]}..{nzl[-1]} ({len(nzl)} distinct) lo/hi element {halves[0]}/{halves[1]}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
