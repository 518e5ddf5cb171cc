"""Experiment: OmnibusTest on a raster far beyond the benchmark's size on ONE GPU -- a large part of
config 5's 24 x 16384 x 16384 stack held in the 288 GB of HBM (plane strides beyond 2^31 bytes,
pixel indices beyond 2^27), checked against the oracle on sampled pixels and rows."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
from oracle import checks
dev = torch.device('cuda:0')
k, ny, nx = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 10240, 16384
t0 = time.time()
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=77, device=dev, change_frac=0.01)
torch.cuda.synchronize()
print('generated %.1f GB in %.1f s' % (st.numel() * 4 / 1e9, time.time() - t0), flush=True)
for alpha in (0.99, 0.01):
    ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res = checks.omnibus_sample(st, ch, alpha, 9, nsample=20000, rows=(0, ny // 2, ny - 1), seed=2)
    print(json.dumps({'raster': '%dt x %d x %d f32' % (k, ny, nx), 'alpha': alpha, 'ms': ms,
                      'Mpx_per_s': ny * nx / ms / 1e3, 'oracle_sample': res}), flush=True)
    del ch
