"""Cliff probe: OmnibusTest parameters off the benchmark's -- number of looks n (the reference's default is 1), tiny series, odd widths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for k, looks in ((24, 9), (24, 1), (24, 4), (48, 1), (12, 1)):
    st = synth.wishart_c2_stack(k, 2048, 4096, looks=looks, seed=1, device=dev, change_frac=0.01)
    for n in sorted({1, looks, 50}):
        for alpha in (0.01, 0.99):
            ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=n))
            ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=n)
            print('c2 f32 k=%d data looks=%d n=%d alpha=%g: %.2f ms  flagged %.3f' % (k, looks, n, alpha, ms, (ch.sum(dim=2) > 0).float().mean().item()), flush=True)
    del st; torch.cuda.empty_cache()
for k in (2, 3, 4, 5):
    st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=1, device=dev, change_frac=0.01)
    for alpha in (0.01, 0.99):
        print('c2 f32 k=%d 2048x4096 alpha=%g: %.2f ms' % (k, alpha, t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))), flush=True)
    del st; torch.cuda.empty_cache()
for nx in (4095, 4097, 1000, 63):
    st = synth.wishart_c2_stack(24, 2048, nx, looks=9, seed=1, device=dev, change_frac=0.01)
    for alpha in (0.01, 0.99):
        ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))
        print('c2 f32 k=24 2048x%d alpha=%g: %.2f ms  (%.3f ns/px)' % (nx, alpha, ms, ms * 1e6 / (2048 * nx)), flush=True)
    del st; torch.cuda.empty_cache()
