"""Round 5: is LDS-DMA the ingredient on the DISTURBER's side of the cross-stream wrong-result class?

Victim: the FFT-form STFT of a library whose FFT part (part 1) is compiled WITH packed-fp32 instructions -- the round-4 reproducer (99-100
of 100 launches wrong next to the 192-row ring conv tile; the product is built without them and is clean).  Disturbers, all the 192-row
ring tile `aero_conv_ring_kernel<2, 4, 3, 3, 0>` on another stream:
    release   as shipped: operands HBM -> LDS by `global_load_lds_dwordx4`
    noglds    the same kernel with every direct copy replaced by global_load_dwordx4 + ds_write_b128 (-DAERO_DBG_NO_GLDS, part 6 only)
plus the nopk victim (the release library) next to both as the control.
usage: disturber_ab.py --build   (no GPU needed)   |   AERO_ALLOW_PACKED_FP32=1 disturber_ab.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
HERE = os.path.join(ROOT, 'tools', 'dbg')


def lib_path(tag):
    return os.path.join(HERE, f'libaero_hip_{tag}.so')


def build():
    import __graft_entry__ as g
    g.build_library()
    g.build_library(out=lib_path('pkvictim'), objdir=os.path.join(HERE, 'build', 'pkvictim'), only_parts=[1], packed_fp32=True)
    g.build_library(out=lib_path('noglds'), objdir=os.path.join(HERE, 'build', 'noglds'), only_parts=[6], defines=['AERO_DBG_NO_GLDS'])


def main():
    if '--build' in sys.argv:
        return build()
    assert os.environ.get('AERO_ALLOW_PACKED_FP32') == '1', 'run with AERO_ALLOW_PACKED_FP32=1 (the victim is an experiment build)'
    import torch
    import concurrency_cases as cc
    from aero_amd import _lib
    from aero_amd.engine import HipEngine
    from conftest import GOLDEN, build_model
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    rounds = int(os.environ.get('PROBE_ITERS', '100'))
    hr = (0.05 * torch.randn(16, 1, 32000, generator=torch.Generator().manual_seed(3))).cuda()
    for vtag in ('pkvictim', 'release'):
        vlib = _lib.load(None if vtag == 'release' else lib_path(vtag))
        m = build_model(meta, 'full').cuda()
        object.__setattr__(m, '_engine', HipEngine(m, lib=vlib))
        m._get_engine().streams = 1
        for dtag in ('release', 'noglds'):
            dlib = _lib.load(None if dtag == 'release' else lib_path(dtag))
            dist = cc.RingDisturber(dlib, 'cuda')
            name = dist.kernel_name()
            # the noglds disturber must still compute the right thing (it is the same kernel through a different copy path)
            ref = cc.RingDisturber(_lib.load(), 'cuda')
            ref.launch(1)
            dist.launch(1)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref.dst, dist.dst))
            bad, first = cc.overlapped(lambda: m._spec(hr, scale=True), dist.launch, rounds, n_disturb=4)
            print(f'victim {vtag:9s} ({vlib.version[:48]}) | disturber {dtag:8s} {name[:44]} (output == release: {same}) | '
                  f'FFT-form STFT wrong in {bad} of {rounds}', flush=True)
            if first:
                print('      ', first[:170], flush=True)


if __name__ == '__main__':
    main()
