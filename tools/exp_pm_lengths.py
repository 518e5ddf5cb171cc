"""Cliff probe: the reference layout (y, x, time) at series lengths that are not a multiple of the 16-byte vector."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for dt in (torch.float32, torch.float64):
    for k in (10, 13, 21, 22, 23, 24):
        st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=1, device=dev, change_frac=0.01).to(dt)
        v = [st[i].permute(1, 2, 0).contiguous() for i in range(4)]
        c12 = torch.complex(v[1], v[2])
        ref = {}
        for alpha in (0.01, 0.99):
            ref[alpha] = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
            tp = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))
            for name, pm in (('split', (v[0], v[1], v[2], v[3])), ('complex', (v[0], c12.real, c12.imag, v[3]))):
                out = kernels.change_detection_pixel_major(*pm, alpha=alpha, n=9)
                if out is None:
                    print('%s k=%d %s alpha=%g: declined (planar: %.2f ms)' % (str(dt)[6:], k, name, alpha, tp), flush=True)
                    continue
                ms = t_ms(lambda: kernels.change_detection_pixel_major(*pm, alpha=alpha, n=9))
                print('%s k=%d %s alpha=%g: %.2f ms (planar %.2f)  equal %s' % (str(dt)[6:], k, name, alpha, ms, tp, bool(torch.equal(out, ref[alpha]))), flush=True)
        del st, v, c12, ref; torch.cuda.empty_cache()
