"""Tuning experiment (GPU box): where OmnibusTest.apply spends its time on a device-resident dataset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import synth, xr_lite, kernels, change as ch, _device
from nd_amd.io import disassemble_complex
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
ds = xr_lite.Dataset()
yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]
ds['C11'] = (('y', 'x', 'time'), yxt[0]); ds['C12'] = (('y', 'x', 'time'), torch.complex(yxt[1], yxt[2])); ds['C22'] = (('y', 'x', 'time'), yxt[3])
del st, yxt
def T(name, fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print('%-28s %.2f ms' % (name, (time.perf_counter() - t0) / n * 1e3)); return r
dm = T('disassemble_complex', lambda: disassemble_complex(ds))
stack = T('_covariance_planes', lambda: ch._covariance_planes(dm, dev))
res = T('kernels.change_detection', lambda: kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=0.99, n=9))
T('.bool()', lambda: res.bool())
T('one transposed copy', lambda: stack[0].copy_(dm['C11'].transpose('time', 'y', 'x').values))
T('one transposed copy (imag of complex)', lambda: stack[2].copy_(dm['C12__im'].transpose('time', 'y', 'x').values))
