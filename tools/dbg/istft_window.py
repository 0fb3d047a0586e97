"""Round 4: WHICH kernel of a forward disturbs the iSTFT?  The round-3 narrowing toggled environment switches, which changes
the forward's whole kernel set; here the forward is left alone and the iSTFT is confined to a WINDOW of it: the victim launch waits for an
event recorded in front of library call i of the forward (other stream) and the forward's call i + w waits for the victim's completion, so
the iSTFT shares the chip only with calls [i, i + w).  Windows are swept over the whole forward; per window: rounds whose waveform differs
from the solo iSTFT in any bit.  usage: istft_window.py [width] [rounds]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class Hook:
    """stands in for the engine's Lib: counts C-ABI calls, fires callbacks in front of chosen call indices, records the kernel names"""

    def __init__(self, lib):
        self._lib = lib
        self.n = 0
        self.at = {}
        self.names = []
        self.record = False

    def __getattr__(self, k):
        return getattr(self._lib, k)

    def call(self, name, *args):
        cb = self.at.get(self.n)
        if cb is not None:
            cb()
        self.n += 1
        self._lib.call(name, *args)
        if self.record:
            self.names.append(self._lib.cdll.aero_last_kernel_name().decode() or name)


def main():
    import torch
    from aero_amd import _lib
    from aero_amd.engine import HipEngine
    from conftest import GOLDEN, build_model
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    lib = _lib.load()
    m = build_model(meta, 'full').cuda()                      # the victim's model (plain library)
    m.eval()
    d = build_model(meta, 'full').cuda()                      # the disturber: same weights, its engine's library calls are hooked
    hook = Hook(lib)
    eng = HipEngine(d, lib=lib)
    eng.ops.lib = hook
    eng.streams = 1
    object.__setattr__(d, '_engine', eng)
    m._get_engine().streams = 1
    x = torch.randn(32, 1, 8000, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        _, s0 = m(x, return_spec=True)
        s16 = s0[:16].contiguous()
        ref = m._ispec(s16).clone()
        hook.record = True
        hook.n = 0
        d(x[16:])
        hook.record = False
        torch.cuda.synchronize()
        ncall = hook.n
        names = list(hook.names)
        print('library calls per forward:', ncall, flush=True)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        total_bad = 0
        for i0 in range(0, ncall, width):
            bad = 0
            for it in range(rounds):
                cur = torch.cuda.current_stream()
                sa.wait_stream(cur)
                sb.wait_stream(cur)
                out = {}
                e_go, e_done = torch.cuda.Event(), torch.cuda.Event()

                def start():
                    e_go.record(sb)
                    with torch.cuda.stream(sa):
                        sa.wait_event(e_go)
                        out['y'] = m._ispec(s16)
                        e_done.record(sa)

                def stop():
                    sb.wait_event(e_done)
                hook.n = 0
                hook.at = {i0: start, min(i0 + width, ncall - 1): stop} if i0 + width < ncall else {i0: start}
                with torch.cuda.stream(sb):
                    d(x[16:])
                torch.cuda.synchronize()
                hook.at = {}
                if not torch.equal(out['y'], ref):
                    bad += 1
            total_bad += bad
            ks = [n.split('(')[0].replace('void ', '') for n in names[i0:i0 + width]]
            print(f'window [{i0:3d}, {i0 + width:3d}): {bad:3d} of {rounds} differ   {" | ".join(k[:44] for k in ks)}', flush=True)
        print('total', total_bad)


if __name__ == '__main__':
    main()
