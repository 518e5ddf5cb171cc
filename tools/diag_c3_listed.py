"""How many pixels does the full-pol streaming search hand to pass B?  (reads the list counters)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import _lib, synth
k, ny, nx = 48, 1024, 8192
dev = torch.device('cuda:0')
st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=4321, device=dev, change_frac=0.01)
L = _lib.lib()
nbytes = L.nd_amd_omnibus_c3_workspace_bytes(ny, nx, k)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
ch = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
planes = (C.c_void_p * 9)(*[st[c].data_ptr() for c in range(9)])
for alpha in (0.01, 1e-4, 0.99):
    rc = L.nd_amd_omnibus_c3(planes, 0, ny, nx, k, st.stride(2), st.stride(3), st.stride(1), 9, alpha,
                             C.c_void_p(ch.data_ptr()), None, None, C.c_void_p(ws.data_ptr()), nbytes,
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    cnt = ws[:128 * 128].view(torch.int32).view(128, 32)[:, 0].sum().item()
    print('alpha', alpha, 'rc', rc, 'listed pixels', cnt, 'of', ny * nx, '= %.4f' % (cnt / (ny * nx)))
