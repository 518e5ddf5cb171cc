"""Time stamps (s_memtime) of one block of the fused multilooking kernel, ND_ML_TRACE build.
    python tools/exp_ml.py build trace=-DND_ML_TRACE       (container)
    ND_AMD_LIB=_variants/libml_trace.so python tools/exp_ml_trace.py [ml] [block]   (GPU box)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 3
blk = sys.argv[2] if len(sys.argv) > 2 else '1000'
dev = torch.device('cuda:0')
tr = torch.zeros(12 * 128 * 12, dtype=torch.int64, device=dev)
os.environ['ND_AMD_ML_TRACE'] = '%x' % tr.data_ptr()
os.environ['ND_AMD_ML_TRACE_BLOCK'] = blk
from nd_amd import kernels, synth
st = synth.wishart_c2_stack(24, 4096, 4096, looks=1, seed=1234, device=dev, change_frac=0.01)
for _ in range(3):
    tr.zero_()
    kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=0.99, ml=ml)
torch.cuda.synchronize()
t = tr.cpu().numpy()[:12 * 16 * 5].reshape(12, 16, 5).astype(np.int64) & 0xffffffff
t0 = t.astype(np.float64)
nst = 16
print('cycles per step (wave 0, steps 32 .. 47 of block %s):' % blk, (t0[0, -1, 0] - t0[0, 0, 0]) / (nst - 1))
names = ['A(carry,res,dma)', 'B(compute)', 'C(wait dma)', 'barrier']
for w in range(12):
    d = np.diff(t0[w], axis=1)
    gap = t0[w, 1:, 0] - t0[w, :-1, 4]            # tail / loop overhead between steps
    print('wave %2d mean cycles:' % w, ' '.join('%s=%.0f' % (n, v) for n, v in zip(names, d.mean(axis=0))),
          'between=%.0f (max %.0f)' % (gap.mean(), gap.max()))
arr = t0[:, :, 3]
print('arrival skew at the barrier: mean max-min = %.0f cycles' % (arr.max(axis=0) - arr.min(axis=0)).mean())
