"""Non-local means timings used while tuning nlmeans.hip: config 3 in both patch modes, the signed
mode with n_eff and with 4 variables, and the tutorial's three-date window."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels

def t_ms(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(7)
res = {}
x = -0.25 * torch.log(torch.rand((1, 12, 4096, 4096, 4), generator=g, device=dev).clamp_min(1e-6)).sum(dim=-1)
y = torch.empty_like(x)
for pm, n in ((1, 3), (0, 10)):
    res['cc_pm%d' % pm] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), y.permute(2, 3, 1, 0), (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1, patch_mode=pm), n)
res['cc_pm1_neff'] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), y.permute(2, 3, 1, 0), (10, 10, 0), (3, 3, 0), 0.5, 0.5, 2.0, patch_mode=1), 2)
for f in (1, 2):
    res['pm1_r5_f%d' % f] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), y.permute(2, 3, 1, 0), (5, 5, 0), (f, f, 0), 0.5, 0.5, -1, patch_mode=1), 3)
chk = float(y.double().sum().item())
del x, y
x4 = torch.rand((4, 6, 2048, 2048), generator=g, device=dev) + 0.5
y4 = torch.empty_like(x4)
res['v2_pm1_r5_f1'] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x4[:2].permute(2, 3, 1, 0), y4[:2].permute(2, 3, 1, 0), (5, 5, 0), (1, 1, 0), 0.5, 0.5, -1, patch_mode=1), 2)
res['v4_pm1_r5_f2_neff'] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x4.permute(2, 3, 1, 0), y4.permute(2, 3, 1, 0), (5, 5, 0), (2, 2, 0), 0.5, 0.5, 2.0, patch_mode=1), 2)
res['v4_pm1_r10_f3'] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(x4.permute(2, 3, 1, 0), y4.permute(2, 3, 1, 0), (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1, patch_mode=1), 2)
xt = torch.rand((4, 24, 1024, 4096), generator=g, device=dev) + 0.5
yt = torch.empty_like(xt)
res['tutorial_r133'] = t_ms(lambda: kernels.pixelwise_nlmeans_3d(xt.permute(1, 2, 3, 0), yt.permute(1, 2, 3, 0), (1, 3, 3), (1, 1, 1), 1.0, 1.0, 50.0, patch_mode=0), 5)
res['checksum'] = chk
print(json.dumps(res))
