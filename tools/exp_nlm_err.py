import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import synth
from nd_amd import tiles, kernels
from oracle import oracle as O
dev = torch.device('cuda:0')
planes = synth.omnibus_stack(seed=12, k=8, ny=60, nx=72, looks=4, dtype=np.float32, change_frac=0.2, factor=5.0)
stack = torch.from_numpy(np.stack(planes)).to(dev)
a = np.ascontiguousarray(np.stack(planes).transpose(2, 3, 1, 0))
for (s, h) in [(0.5, 2.0), (0.5, 0.5), (0.1, 0.2), (1.0, 1.0)]:
    want = np.empty_like(a)
    O.pixelwise_nlmeans_3d(a, want, (3, 3, 0), (1, 1, 0), s, h, -1, njobs=8, patch_mode=1)
    full_f = tiles.nlmeans_rows(stack, 60, (0, 3, 3), (0, 1, 1), s, h, patch_mode=1)
    got = full_f.permute(2, 3, 1, 0).cpu().numpy()
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    os.environ['ND_AMD_NO_TILED'] = '1'
    print('sigma', s, 'h', h, 'max rel', rel.max(), 'p99.9', np.quantile(rel, 0.999), 'maxabs', np.abs(got - want).max(), 'where want=', want.flat[rel.argmax()])
