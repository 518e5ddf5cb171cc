"""Experiment: look for performance cliffs of the omnibus entry points over (k, alpha, dtype, stats)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
what = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if what == 'c2':
    for dt, ks in ((torch.float32, (24, 96)), (torch.float64, (12, 16, 24, 32, 48))):
        for k in ks:
            ny, nx = 2048, 4096
            st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1, device=dev, change_frac=0.01).to(dt)
            for alpha in (1e-4, 0.01, 0.1, 0.5, 0.99):
                for stats in (False, True):
                    if dt == torch.float32 and k == 24 and not stats: continue
                    ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9, stats=stats), 2)
                    print('c2 %s k=%d 2048x4096 alpha=%g stats=%d: %.2f ms' % (str(dt)[6:], k, alpha, stats, ms), flush=True)
            del st; torch.cuda.empty_cache()
else:
    k, ny, nx = 48, 512, 4096
    st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=2, device=dev, change_frac=0.01)
    for alpha in (0.99, 0.5, 0.01):
        ms = t_ms(lambda: kernels.change_detection_c3([st[c] for c in range(9)], alpha=alpha, n=9), 1)
        print('c3 f32 k=48 512x4096 alpha=%g: %.2f ms' % (alpha, ms), flush=True)
