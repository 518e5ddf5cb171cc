"""Secondary measurement: the tutorial pipeline of BASELINE config 5 on one GPU's row block:
NLMeansFilter(dims=('time','y','x'), r=(1,3,3), f=(1,1,1), n_eff=50-like) -> OmnibusTest."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, synth, tiles
ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=24); ap.add_argument('--ny', type=int, default=2048); ap.add_argument('--nx', type=int, default=4096)
ap.add_argument('--n-eff', type=float, default=50.0); ap.add_argument('--patch-mode', type=int, default=0)
ap.add_argument('--steps', type=int, default=2); ap.add_argument('--alpha', type=float, default=0.99)
a = ap.parse_args()
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(a.k, a.ny, a.nx, looks=9, seed=5, device=dev, change_frac=0.01)
def run():
    return tiles.nlmeans_then_omnibus(st, a.ny, (1, 3, 3), (1, 1, 1), 0.5, 0.5, a.alpha, 9, n_eff=a.n_eff, patch_mode=a.patch_mode)
out = run(); torch.cuda.synchronize()
_lib.timing_enable(64)
t0 = time.perf_counter()
for _ in range(a.steps): out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
kt = _lib.timing_collect(); by = {}
for n_, ms in kt: by.setdefault(n_, []).append(ms)
print(json.dumps({'workload': 'nlmeans(time,y,x r=(1,3,3) f=(1,1,1) n_eff=%g, patch_mode %d) -> omnibus, %dt x %d x %d f32 x 4 vars' % (a.n_eff, a.patch_mode, a.k, a.ny, a.nx),
                  'alpha': a.alpha, 'ms': dt * 1e3, 'Mpx_per_s': a.ny * a.nx / dt / 1e6, 'kernels_ms': {n_: sum(v) / len(v) for n_, v in by.items()},
                  'flagged': float((out.sum(dim=2) > 0).float().mean().item())}))
