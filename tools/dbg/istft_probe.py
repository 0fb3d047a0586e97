"""Round-4 experiments on the open iSTFT finding.  The iSTFT runs on one stream while a second stream runs (a) a whole
forward, as in round 3, or (b) only the 192-row ring conv tile; every result is compared bit for bit with the solo run.
With tools/dbg/libaero_hip_dbg.so (built by `python tools/dbg/istft_probe.py --build`, -DAERO_ISTFT_DEBUG) the kernel
  mode 1: re-reads every spectrum value past the caches (sc0 sc1) and counts the values that differ from its ordinary load;
  mode 2: takes the cache-bypassing loads as THE loads.
If mode 1 counts mismatches where the output is wrong, the ordinary load returned wrong data (vector L1 / L2 side); if the output goes
wrong with zero mismatches, the value was damaged after it arrived (registers / LDS).  If mode 2 never fails, bypassing is the fix."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
DBG = os.path.join(ROOT, 'tools', 'dbg', 'libaero_hip_dbg.so')


def build():
    import __graft_entry__ as g
    g.build_library(out=DBG, defines=['AERO_ISTFT_DEBUG'], objdir=os.path.join(ROOT, 'tools', 'dbg', 'build'))


def main():
    if '--build' in sys.argv:
        return build()
    import torch
    import concurrency_cases as cc
    from aero_amd import _lib
    from aero_amd.engine import HipEngine
    from conftest import GOLDEN, build_model
    iters = int(os.environ.get('PROBE_ITERS', '300'))
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    x = torch.randn(32, 1, 8000, generator=torch.Generator().manual_seed(5)).cuda()
    for tag, path in (('release', None), ('debug', DBG)):
        if path and not os.path.exists(path):
            print(tag, 'library missing:', path)
            continue
        lib = _lib.load(path)
        if path:
            import ctypes as C
            lib.cdll.aero_istft_debug_set.argtypes = [C.c_void_p, C.c_int32]
            lib.cdll.aero_istft_debug_set.restype = C.c_int32
        m = build_model(meta, 'full').cuda()
        object.__setattr__(m, '_engine', HipEngine(m, lib=lib))
        eng = m._get_engine()
        eng.streams = 1
        dist = cc.RingDisturber(lib, 'cuda')
        print(tag, 'disturber kernel:', dist.kernel_name(), flush=True)
        with torch.no_grad():
            _, s0 = m(x, return_spec=True)
        s16 = s0[:16].contiguous()
        torch.cuda.synchronize()
        fwd = lambda n: [m(x[16:]) for _ in range(max(1, n // 24))]      # noqa: E731  (a whole forward of other clips, as in round 3)
        modes = [0] if path is None else [0, 1, 2]
        for mode in modes:
            cnt = torch.zeros(8, dtype=torch.int64, device='cuda')
            if path:
                lib.cdll.aero_istft_debug_set(cnt.data_ptr() if mode & 1 else None, mode)
            for dname, dfn, nd in (('forward', fwd, 24), ('ring192', dist.launch, 6)):
                cnt.zero_()
                bad, first = cc.overlapped(lambda: m._ispec(s16), dfn, iters, n_disturb=nd)
                c = cnt.tolist()
                print(f'{tag} mode {mode} istft next to {dname}: {bad} of {iters} rounds differ; load mismatches {c[0]}'
                      + (f' first: addr {c[1]:#x} got {c[2]:#018x} memory {c[3]:#018x} blk/thread {c[4]:#x} bin/frame {c[5]:#x}' if c[0] else '')
                      + (f' | {first}' if first else ''), flush=True)
        if path:
            lib.cdll.aero_istft_debug_set(None, 0)
        # the other families next to the ring tile: the forward without its iSTFT (spectrogram outputs only)
        bad, first = cc.overlapped(lambda: m(x[:8], return_spec=True, return_lr_spec=True)[1:], dist.launch, 60, n_disturb=40)
        print(f'{tag} forward (spectrogram outputs) next to ring192: {bad} of 60 rounds differ {first}', flush=True)


if __name__ == '__main__':
    main()
