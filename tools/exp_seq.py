import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(24, 4096, 4096, seed=1234, device=dev, change_frac=0.01)
def run(n):
    for _ in range(n):
        out = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    return out
run(3)
_lib.timing_enable(256)
torch.cuda.synchronize(); time.sleep(0.05)
t0 = time.perf_counter(); run(50); torch.cuda.synchronize(); dt = time.perf_counter() - t0
kt = _lib.timing_collect()
a = [round(ms, 3) for n, ms in kt if n == 'omnibus_c2_global']
b = [round(ms, 3) for n, ms in kt if n == 'omnibus_c2_search']
print('wall per step ms', dt / 50 * 1e3)
print('A', a)
print('B', b)
