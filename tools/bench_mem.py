"""Calibration: achievable HBM rates on this GPU for plain torch fill / copy / reduce (reference points for the
HBM-bound kernels of the path).  python tools/bench_mem.py"""
import torch


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    for mb in (25, 100, 200, 400, 800):
        n = mb * 1024 * 1024 // 2
        x = torch.randn(n, device='cuda').half()
        y = torch.empty_like(x)
        t_fill = timeit(lambda: y.fill_(1.0))
        t_copy = timeit(lambda: y.copy_(x))
        t_read = timeit(lambda: x.view(torch.int16).max())
        t_add = timeit(lambda: torch.add(x, 1.0, out=y))
        print(f'{mb:4d} MB: fill {mb / t_fill / 1e3 * 1.048576:5.2f} TB/s ({t_fill * 1e3:6.1f} us)  copy(r+w) {2 * mb / t_copy / 1e3 * 1.048576:5.2f} TB/s '
              f'({t_copy * 1e3:6.1f} us)  max-reduce {mb / t_read / 1e3 * 1.048576:5.2f} TB/s ({t_read * 1e3:6.1f} us)  add(r+w) {2 * mb / t_add / 1e3 * 1.048576:5.2f} TB/s',
              flush=True)


if __name__ == '__main__':
    main()
