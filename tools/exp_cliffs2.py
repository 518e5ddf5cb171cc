import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nd_amd import kernels
dev = torch.device('cuda:0')
def t_ms(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
g = torch.Generator(device=dev).manual_seed(3)
x = torch.rand((8, 2048, 2048), generator=g, device=dev) + 0.5
y = torch.empty_like(x)
for w in (15, 17, 21, 31):
    kern = np.ones((1, w, w)) / (w * w)
    print('conv f32 box%d 8x2048x2048: %.2f ms' % (w, t_ms(lambda: kernels.convolve(x, kern, out=y))), flush=True)
rk = np.random.default_rng(0).normal(size=(1, 17, 17))
print('conv f32 rand17 8x2048x2048: %.2f ms' % t_ms(lambda: kernels.convolve(x, rk, out=y)), flush=True)
for sg in (3.0, 5.0, 10.0):
    print('gauss f32 sigma=%g: %.2f ms' % (sg, t_ms(lambda: kernels.gaussian_filter(x, (0, sg, sg), out=y))), flush=True)
del x, y
for nv, k_, ny, nx in ((1, 4, 1024, 2048),):
    a = (torch.rand((nv, k_, ny, nx), generator=g, device=dev) + 0.5)
    o = torch.empty_like(a)
    for pm in (0, 1):
        for r, f in (((0, 12, 12), (0, 3, 3)), ((0, 15, 15), (0, 3, 3)), ((0, 5, 5), (0, 4, 4)), ((0, 10, 10), (0, 5, 5)), ((0, 3, 5), (0, 1, 2)), ((0, 4, 4), (0, 2, 2))):
            def run():
                kernels.pixelwise_nlmeans_3d(a.permute(2, 3, 1, 0), o.permute(2, 3, 1, 0), (r[1], r[2], 0), (f[1], f[2], 0), 0.5, 0.5, -1, patch_mode=pm, neff_policy=0)
            ms = t_ms(run, 1)
            nq = (2 * r[1] + 1) * (2 * r[2] + 1) - 1
            print('nlm f32 nv=%d pm=%d r=%s f=%s on %dx%dx%d: %.2f ms  %.4f ns/(elem.neighbour)' % (nv, pm, r, f, k_, ny, nx, ms, ms * 1e6 / (a.numel() * nq)), flush=True)
for nv in (5, 8):
    a = (torch.rand((nv, 4, 512, 1024), generator=g, device=dev) + 0.5)
    o = torch.empty_like(a)
    for pm in (0, 1):
        def run():
            kernels.pixelwise_nlmeans_3d(a.permute(2, 3, 1, 0), o.permute(2, 3, 1, 0), (3, 3, 0), (1, 1, 0), 0.5, 0.5, -1, patch_mode=pm, neff_policy=0)
        print('nlm f32 nv=%d pm=%d r=3 f=1 on 4x512x1024: %.2f ms' % (nv, pm, t_ms(run, 1)), flush=True)
