import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nd_amd import kernels, synth
from oracle import oracle as O
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
st = synth.wishart_c3_stack(24, 512, 4096, looks=9, seed=2, device=dev, change_frac=0.01)
pl = [st[c] for c in range(9)]
for n in (1, 9):
    for alpha in (0.01, 0.99):
        ms = t_ms(lambda: kernels.change_detection_c3(pl, alpha=alpha, n=n))
        ch = kernels.change_detection_c3(pl, alpha=alpha, n=n)
        print('c3 f32 k=24 512x4096 n=%d alpha=%g: %.2f ms  changes %d' % (n, alpha, ms, int(ch.sum().item())), flush=True)
# oracle check, small
for k, n in ((5, 1), (24, 1), (24, 2), (12, 3)):
    st2 = synth.wishart_c3_stack(k, 16, 130, looks=9, seed=20 + k, device=dev, change_frac=0.3)
    yxt = [np.ascontiguousarray(st2[c].permute(1, 2, 0).cpu().numpy()) for c in range(9)]
    for alpha in (0.01, 0.5, 0.99):
        want = O.change_detection_pol(yxt, 3, alpha, n, njobs=8)
        got = kernels.change_detection_c3([st2[c] for c in range(9)], alpha=alpha, n=n).cpu().numpy()
        print('  k=%d n=%d alpha=%g equal %s (changes %d)' % (k, n, alpha, bool(np.array_equal(got, want)), int(want.sum())), flush=True)
