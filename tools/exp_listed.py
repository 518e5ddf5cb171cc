"""How many pixels the fused (streaming) search hands to the exact pass B, per threshold: the sum
of the shard counters of the candidate list in the workspace after one call through the C-ABI."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
k, ny, nx = int(os.environ.get('K', 24)), int(os.environ.get('NY', 4096)), int(os.environ.get('NX', 4096))
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
L = _lib.lib()
for alpha in [float(x) for x in os.environ.get('ALPHAS', '1e-4,0.01,0.05').split(',')]:
    change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
    nbytes = L.nd_amd_omnibus_c2_workspace_bytes(kernels._DT[st.dtype], ny, nx, k, None)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    p = st[0]
    _lib.check(L.nd_amd_omnibus_c2(kernels._ptr(st[0]), kernels._ptr(st[1]), kernels._ptr(st[2]), kernels._ptr(st[3]),
                                   kernels._DT[st.dtype], ny, nx, k, p.stride(1), p.stride(2), p.stride(0), 9, alpha,
                                   kernels._ptr(change), None, None, kernels._ptr(ws), nbytes, kernels._stream_ptr(dev)))
    torch.cuda.synchronize()
    per = ws[:128 * 32 * 4].view(torch.int32).view(128, 32)[:, 0]
    cnt = per.sum().item()
    print(json.dumps({'alpha': alpha, 'listed': int(cnt), 'fraction': cnt / (ny * nx), 'shard_max': int(per.max().item()), 'shard_min': int(per.min().item()),
                      'marked_for_exact': int(ws[:128 * 32 * 4].view(torch.int32)[3].item())}))
