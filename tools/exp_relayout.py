"""Tuning experiment (GPU box): the (y, x, time) -> planar transpose kernel against torch's permuted copy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth, _lib
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
src = torch.rand((ny, nx, k), device=dev)
c = torch.complex(src, src * 2)
stack = synth.empty_stack(2, k, ny, nx, dev)
def T(name, fn, n=5):
    r = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print('%-34s %.3f ms  -> %s' % (name, (time.perf_counter() - t0) / n * 1e3, r))
T('relayout real', lambda: kernels.relayout_planar(src, stack[0]))
T('relayout imag half', lambda: kernels.relayout_planar(c.imag, stack[1]))
T('torch permuted copy', lambda: stack[0].copy_(src.permute(2, 0, 1)) is None)
print(torch.equal(stack[1], c.imag.permute(2, 0, 1)))
