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
run(5)
_lib.timing_enable(1024)
torch.cuda.synchronize()
t0 = time.perf_counter(); out = run(50); torch.cuda.synchronize(); dt = time.perf_counter() - t0
kt = _lib.timing_collect()
a = sum(ms for n, ms in kt if n == 'omnibus_c2_global') / 50
b = sum(ms for n, ms in kt if n == 'omnibus_c2_search') / 50
print('chunks', os.environ.get('ND_AMD_OMNIBUS_CHUNKS'), 'wall per step ms %.4f' % (dt / 50 * 1e3), 'sumA %.4f sumB %.4f' % (a, b), 'changes', int(out.sum().item()))
