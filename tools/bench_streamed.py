"""PCIe-inclusive rate: OmnibusTest on a HOST-resident stack through nd_amd.streaming."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from nd_amd import streaming, synth
k, ny, nx = 24, 4096, 4096
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(k, ny, nx, seed=1234, device=dev, change_frac=0.01)
host = [st[v].cpu().numpy() for v in range(4)]
del st
out = np.empty((ny, nx, k), np.uint8)
for rows in (256, 512, 1024):
    streaming.omnibus_streamed(host, 0.99, 9, rows_per_tile=rows, out=out)
    t0 = time.perf_counter()
    streaming.omnibus_streamed(host, 0.99, 9, rows_per_tile=rows, out=out)
    dt = time.perf_counter() - t0
    print(json.dumps({'workload': 'host-resident 24t x 4096 x 4096 f32, rows_per_tile=%d' % rows, 's': dt,
                      'Mpx_per_s': ny * nx / dt / 1e6, 'GBps_in': 4 * k * ny * nx * 4 / dt / 1e9,
                      'changes': int(out.sum())}))
