"""Where the ring-resident iSTFT (aero_istft2_kernel) spends its time: a build of the FFT part of the library with -DAERO_ISTFT_ABLATION
(the release library has no such switch) leaves out one phase at a time -- AERO_ISTFT_ABL bits: 1 spectrum loads of groups > 0,
2 frame transforms, 4 overlap-add, 8 unpack / deposit.  Results are wrong by construction; only the time is read.
usage: istft_ablation.py --build (here) | istft_ablation.py (MI355X)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.join(ROOT, 'tools', 'dbg')
LIB = os.path.join(HERE, 'libaero_hip_istft_abl.so')


def one():
    import torch
    from aero_amd import _lib
    lib = _lib.load(LIB)
    B, n_fft, hop, T = 64, 512, 64, 501
    spec = torch.randn(B, n_fft // 2, T, 2, device='cuda')
    win = torch.hann_window(n_fft, device='cuda')
    env = torch.ones(n_fft + hop * (T - 1), device='cuda')
    Lout = hop * (T - 1)
    y = torch.empty(B, Lout, device='cuda')
    s = torch.cuda.current_stream().cuda_stream

    def go():
        lib.call('aero_istft_fwd', spec.data_ptr(), B, n_fft // 2, T, n_fft, hop, win.data_ptr(), env.data_ptr(), y.data_ptr(), Lout, s)
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"ABL={os.environ.get('AERO_ISTFT_ABL', '0'):>3s}: {best * 1e3:7.1f} us", flush=True)


def main():
    if '--build' in sys.argv:
        import __graft_entry__ as g
        g.build_library()
        return g.build_library(out=LIB, objdir=os.path.join(HERE, 'build', 'istft_abl'), only_parts=[1], defines=['AERO_ISTFT_ABLATION'])
    if '--one' in sys.argv:
        return one()
    for abl in (0, 1, 2, 4, 8, 3, 6, 7, 14, 15):
        env = dict(os.environ, AERO_ISTFT_ABL=str(abl))
        subprocess.run([sys.executable, os.path.abspath(__file__), '--one'], env=env, check=False)


if __name__ == '__main__':
    main()
