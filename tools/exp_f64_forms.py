import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for k in (32, 48):
    st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=1, device=dev, change_frac=0.01).to(torch.float64)
    for alpha in (0.01, 0.05, 0.1, 0.5):
        ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))
        print('f64 k=%d alpha=%g form=%s: %.2f ms' % (k, alpha, os.environ.get('ND_AMD_FUSED_FORM', 'default'), ms), flush=True)
    del st; torch.cuda.empty_cache()
