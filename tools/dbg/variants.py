"""Round 4: the FFT-form STFT comes out wrong (real parts of 16 consecutive bins of a frame, lanes 48-63 of a wave) in ~100 % of the
launches that share the chip with the 192-row ring conv tile, and the iSTFT in a few % of those that overlap certain MFMA kernels of a
forward.  Experiment builds of the FFT part of the library (part 1: k_stft.h + k_train.h; every other part is the release object):
    nopk   compiled without packed-fp32 instructions (-target-feature -packed-fp32-ops): 4854 v_pk_{fma,mul,add}_f32 -> 0
    zlds   the STFT zeroes its whole LDS allocation first (stale LDS contents of the previous workgroup on that CU)
    wsync  the wave-level rendezvous between LDS passes also waits lgkmcnt(0) (instead of relying on in-order LDS execution)
usage: variants.py --build | variants.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
HERE = os.path.join(ROOT, 'tools', 'dbg')
VARIANTS = {'nopk': dict(flags=['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']),
            'zlds': dict(defines=['AERO_DBG_ZERO_LDS']),
            'wsync': dict(defines=['AERO_DBG_WAVE_SYNC_WAIT'])}


def lib_path(tag):
    return os.path.join(HERE, f'libaero_hip_{tag}.so')


def build():
    import __graft_entry__ as g
    g.build_library()                                            # the release objects the variants link against
    for tag, kw in VARIANTS.items():
        g.build_library(out=lib_path(tag), objdir=os.path.join(HERE, 'build', tag), only_parts=[1], **kw)



def build_nopk_all():
    """the whole library without packed-fp32 instructions (a candidate for the release build: measured against it in the bench)"""
    import __graft_entry__ as g
    g.build_library(out=lib_path('nopk_all'), objdir=os.path.join(HERE, 'build', 'nopk_all'), flags=VARIANTS['nopk']['flags'])


def main():
    if '--build-nopk-all' in sys.argv:
        return build_nopk_all()
    if '--build' in sys.argv:
        return build()
    import torch
    import concurrency_cases as cc
    from aero_amd import _lib
    from aero_amd.engine import HipEngine
    from conftest import GOLDEN, build_model
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    rounds = int(os.environ.get('PROBE_ITERS', '100'))
    hr = (0.05 * torch.randn(16, 1, 32000, generator=torch.Generator().manual_seed(3))).cuda()
    x = torch.randn(32, 1, 8000, generator=torch.Generator().manual_seed(5)).cuda()
    for tag in ['release'] + list(VARIANTS):
        path = None if tag == 'release' else lib_path(tag)
        if path and not os.path.exists(path):
            print(tag, 'missing')
            continue
        lib = _lib.load(path)
        m = build_model(meta, 'full').cuda()
        object.__setattr__(m, '_engine', HipEngine(m, lib=lib))
        m._get_engine().streams = 1
        dist = cc.RingDisturber(_lib.load(), 'cuda')               # the disturber always runs on the RELEASE library
        with torch.no_grad():
            _, s0 = m(x, return_spec=True)
        s16 = s0[:16].contiguous()
        mrel = build_model(meta, 'full').cuda()
        mrel._get_engine().streams = 1
        fwd = lambda n: mrel(x[16:])                               # noqa: E731  (a whole release-library forward of other clips)
        b1, f1 = cc.overlapped(lambda: m._spec(hr, scale=True), dist.launch, rounds, n_disturb=4)
        b2, f2 = cc.overlapped(lambda: m._ispec(s16), fwd, 2 * rounds, n_disturb=1)
        b3, f3 = cc.overlapped(lambda: m._ispec(s16), dist.launch, rounds, n_disturb=6)
        print(f'{tag:8s} FFT-form STFT next to ring192: {b1} of {rounds} | iSTFT next to a forward: {b2} of {2 * rounds} | iSTFT next to ring192: {b3} of {rounds}', flush=True)
        if f1:
            print('          ', f1[:160])


if __name__ == '__main__':
    main()

