"""debug: the sparse regime on a raster whose date stride passes 2^31 bytes -- halves against the whole,
with the change map and workspace landing on recycled (non-zero) memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd_amd import synth, tiles, kernels
dev = torch.device('cuda:0')
k, ny, nx = 24, 2048, 16384
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=55, device=dev, change_frac=0.01)
torch.cuda.synchronize()


def dirty():
    a = torch.full((ny * nx * k,), 0xAB, dtype=torch.uint8, device=dev)
    b = torch.full((2 * ny * nx * k,), 0xCD, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    del a, b


for alpha in (0.99, 0.5, 0.99):
    parts = []
    for r0 in range(0, ny, 512):
        sub = [st[v][:, r0:r0 + 512].contiguous() for v in range(4)]
        parts.append(kernels.change_detection(sub[0], sub[1], sub[2], sub[3], alpha=alpha, n=9, dims=('time', 'y', 'x')))
    ref = torch.cat(parts, dim=0)
    del parts
    dirty()
    whole = tiles.omnibus_rows(st, alpha, 9)
    torch.cuda.synchronize()
    diff = (whole != ref)
    print('alpha', alpha, 'shape', tuple(whole.shape), 'differing bytes', int(diff.sum()), 'of', diff.numel(),
          'nonzero whole/ref', int((whole != 0).sum()), int((ref != 0).sum()), flush=True)
    if diff.any():
        rows = diff.any(dim=2).any(dim=1).nonzero().flatten()
        print(' rows differing: first', int(rows[0]), 'last', int(rows[-1]), 'count', int(rows.numel()))
        cols = diff.any(dim=2).any(dim=0).nonzero().flatten()
        print(' cols differing: first', int(cols[0]), 'last', int(cols[-1]), 'count', int(cols.numel()))
        print(' values in whole where differing:', torch.unique(whole[diff])[:10].tolist())
        dpx = diff.any(dim=2)
        r = int(rows[0])
        cc = dpx[r].nonzero().flatten()
        print(' first differing row', r, 'cols', cc[:8].tolist(), '...', cc[-4:].tolist(), 'n', int(cc.numel()))
        print(' dates differing at first px:', diff[r, int(cc[0])].nonzero().flatten().tolist())
    del whole, ref, diff
