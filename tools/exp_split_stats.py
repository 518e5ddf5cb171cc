import os, sys, time
sys.path.insert(0, os.getcwd())
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
for dt, k, ny in ((torch.float32, 96, 2048), (torch.float32, 64, 2048), (torch.float32, 160, 1024), (torch.float64, 48, 1024), (torch.float64, 96, 1024)):
    st = synth.wishart_c2_stack(k, ny, 4096, looks=9, seed=1, device=dev, change_frac=0.01, dtype=dt)
    a = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9))
    b = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9, stats=True))
    ch, z, P = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9, stats=True)
    print('%s: %s k=%d %dx4096 alpha=0.99: map %.3f ms, with rasters %.3f ms (+%.0f %%)  zsum %.9g Psum %.9g changes %d' % (tag, str(dt)[6:], k, ny, a, b, 100 * (b / a - 1), z.double().nansum().item(), P.double().nansum().item(), int(ch.sum().item())), flush=True)
    del st; torch.cuda.empty_cache()
