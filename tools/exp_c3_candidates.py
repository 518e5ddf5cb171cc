import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, ctypes as C
from nd_amd import synth, _lib
dev = torch.device('cuda:0')
for cyc in (0, 6):
    st = synth.wishart_c3_stack(48, 1024, 8192, looks=9, seed=4321, device=dev, change_frac=0.01, cycle=cyc)
    planes = [st[c] for c in range(9)]
    L = _lib.lib(); ny, nx, k = 1024, 8192, 48
    nbytes = L.nd_amd_omnibus_c3_workspace_bytes(ny, nx, k)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    ch = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in planes])
    _lib.check(L.nd_amd_omnibus_c3(ptrs, 0, ny, nx, k, planes[0].stride(1), planes[0].stride(2), planes[0].stride(0), 9, 0.99, ch.data_ptr(), None, None, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    counts = ws[:128 * 32 * 4].view(torch.int32).view(128, 32)[:, 0]
    print('cycle', cyc, 'candidates', int(counts.sum()), 'max per shard', int(counts.max()), 'dump cap per shard', (ny*nx//16+127)//128, 'changes', int(ch.sum()))
    del st, planes, ws, ch
