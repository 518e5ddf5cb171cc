"""What the z / P rasters cost on top of the map: 2048 x 4096, one build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
tag = os.environ.get('TAG', 'as built')
for dt, k in ((torch.float32, 24), (torch.float64, 24), (torch.float64, 12), (torch.float32, 96)):
    st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=1, device=dev, change_frac=0.01, dtype=dt)
    for alpha in (1e-4, 0.01, 0.5, 0.99):
        a = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))
        b = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9, stats=True))
        print('%s: %s k=%d alpha=%g: map %.3f ms, with rasters %.3f ms (+%.0f %%)' % (tag, str(dt)[6:], k, alpha, a, b, 100 * (b / a - 1)), flush=True)
    del st; torch.cuda.empty_cache()
