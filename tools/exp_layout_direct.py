"""Tuning experiment (GPU box): OmnibusTest kernels fed the reference's (y, x, time) layout directly
(generic-stride path) against transpose + planar path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]
def T(name, fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print('%-44s %.2f ms' % (name, (time.perf_counter() - t0) / n * 1e3)); return r
a = T('planar', lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9))
b = T('(y, x, time) layout, generic strides', lambda: kernels.change_detection(*yxt, alpha=0.99, n=9, dims=('y', 'x', 'time')))
print(torch.equal(a, b))
