"""Config 5's share through the filter with signed patch distances (patch_mode 1, n_eff = -1): time per call and the
check of tests/test_config_share_gpu.py::test_config5_signed_patch_distances."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import synth, tiles, _lib
from oracle import checks
dev = torch.device('cuda:0')
k, ny, nx = 24, 2048, 16384
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=55, device=dev, change_frac=0.01)
r, f = (1, 3, 3), (1, 1, 1)
filt = tiles.nlmeans_rows(st, ny, r, f, 1.0, 1.0, n_eff=-1, patch_mode=1)
torch.cuda.synchronize()
_lib.timing_enable(8)
t0 = time.perf_counter()
filt = tiles.nlmeans_rows(st, ny, r, f, 1.0, 1.0, n_eff=-1, patch_mode=1)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
km = dict(_lib.timing_collect())
res = checks.nlmeans_crops(st, filt, r, f, 1.0, 1.0, -1, 1, [(0, 0), (0, 8000), (1000, 16384), (2048, 16384)], size=(8, 64))
print(json.dumps({'workload': 'NLMeansFilter r=(1,3,3) f=1 signed patch distances on 24 x 2048 x 16384 x 4 (config 5 share)',
                  'ms': round(ms, 2), 'kernels_ms': {n_: round(v, 2) for n_, v in km.items()}, 'check': res}))
