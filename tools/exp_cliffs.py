"""Experiment: look for performance cliffs of the omnibus entry points over (k, alpha, dtype, stats)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
what = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if what == 'c2':
    for dt, ks in ((torch.float32, (24, 96)), (torch.float64, (12, 16, 24, 32, 48))):
        for k in ks:
            ny, nx = 2048, 4096
            st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1, device=dev, change_frac=0.01).to(dt)
            for alpha in (1e-4, 0.01, 0.1, 0.5, 0.99):
                for stats in (False, True):
                    if dt == torch.float32 and k == 24 and not stats: continue
                    ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9, stats=stats), 2)
                    print('c2 %s k=%d 2048x4096 alpha=%g stats=%d: %.2f ms' % (str(dt)[6:], k, alpha, stats, ms), flush=True)
            del st; torch.cuda.empty_cache()
if what == 'c3':
    k, ny, nx = 48, 512, 4096
    st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=2, device=dev, change_frac=0.01)
    for alpha in (0.99, 0.5, 0.01):
        ms = t_ms(lambda: kernels.change_detection_c3([st[c] for c in range(9)], alpha=alpha, n=9), 1)
        print('c3 f32 k=48 512x4096 alpha=%g: %.2f ms' % (alpha, ms), flush=True)

if what == 'filters':
    import numpy as np
    g = torch.Generator(device=dev).manual_seed(3)
    k, ny, nx = 8, 2048, 2048
    for dt in (torch.float32, torch.float64):
        x = (torch.rand((k, ny, nx), generator=g, device=dev) + 0.5).to(dt)
        y = torch.empty_like(x)
        for desc, kern, kw in (('box3', np.ones((1, 3, 3)) / 9, {}), ('box5', np.ones((1, 5, 5)) / 25, {}),
                               ('box9', np.ones((1, 9, 9)) / 81, {}), ('box15', np.ones((1, 15, 15)) / 225, {}),
                               ('rand5 zeros', np.where(np.random.default_rng(0).random((1, 5, 5)) < 0.3, 0, 1.0), {}),
                               ('rand4x6', np.random.default_rng(1).normal(size=(1, 4, 6)), {}),
                               ('box5 constant', np.ones((1, 5, 5)) / 25, dict(mode='constant')),
                               ('box3x3x3', np.ones((3, 3, 3)) / 27, {}), ('box 1x1x7', np.ones((1, 1, 7)) / 7, {}),
                               ('box 3x1x1 (time)', np.ones((3, 1, 1)) / 3, {})):
            ms = t_ms(lambda: kernels.convolve(x, kern, out=y, **kw), 2)
            print('conv %s %-18s 8x2048x2048: %8.2f ms  %.2f ns/elem' % (str(dt)[6:], desc, ms, ms * 1e6 / x.numel()), flush=True)
        for sg in ((0, 1, 1), (0, 2.5, 2.5), (1, 1, 1), (0, 1, 0), (0, 0, 3)):
            ms = t_ms(lambda: kernels.gaussian_filter(x, sg, out=y), 2)
            print('gauss %s sigma=%-14s 8x2048x2048: %8.2f ms' % (str(dt)[6:], sg, ms), flush=True)
    del x, y
    torch.cuda.empty_cache()
if what == 'nlm':
    g = torch.Generator(device=dev).manual_seed(4)
    for dt in (torch.float32, torch.float64):
        for nv in (1, 4):
            k, ny, nx = 6, 1024, 2048
            x = (torch.rand((nv, k, ny, nx), generator=g, device=dev) + 0.5).to(dt)
            y = torch.empty_like(x)
            for pm in (0, 1):
                # (n_eff = 30 with the 48 neighbours of r = 3: uniform data give unit weights, W^2 / W2 = 48, and an
                #  n_eff of 50 sits 2 % from the no-solution boundary -- every pixel then takes the exact path by design)
                for (r, f, ne) in (((0, 3, 3), (0, 1, 1), -1), ((0, 3, 3), (0, 1, 1), 30.0), ((0, 10, 10), (0, 3, 3), -1),
                                   ((1, 3, 3), (1, 1, 1), 50.0), ((2, 3, 3), (1, 1, 1), -1), ((0, 3, 3), (0, 0, 0), -1),
                                   ((0, 6, 6), (0, 2, 2), -1), ((1, 3, 3), (0, 1, 1), -1)):
                    if dt == torch.float64 and (r[1] > 3 or r[0] > 1): continue
                    def run():
                        if r[0] == 0 and f[0] == 0:
                            kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), y.permute(2, 3, 1, 0), (r[1], r[2], 0), (f[1], f[2], 0), 0.5, 0.5, ne, patch_mode=pm, neff_policy=0)
                        else:
                            kernels.pixelwise_nlmeans_3d(x.permute(1, 2, 3, 0), y.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne, patch_mode=pm, neff_policy=0)
                    ms = t_ms(run, 1)
                    nq = (2 * r[0] + 1) * (2 * r[1] + 1) * (2 * r[2] + 1) - 1
                    print('nlm %s nv=%d pm=%d r=%s f=%s n_eff=%g: %9.2f ms  %.3f ns/(elem.neighbour)' % (str(dt)[6:], nv, pm, r, f, ne, ms, ms * 1e6 / (x.numel() * nq)), flush=True)
            del x, y
            torch.cuda.empty_cache()

if what == 'api':
    # the drop-in classes on device datasets: reference layout (y, x, time) and time-first
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    from nd_amd.filters import BoxcarFilter, GaussianFilter, NLMeansFilter, ConvolutionFilter
    import numpy as np
    for k in (24, 48):
        ny, nx = 2048, 4096
        st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=12, device=dev, change_frac=0.01)
        for layout in (('y', 'x', 'time'), ('time', 'y', 'x')):
            ds = xr_lite.Dataset()
            if layout[0] == 'y':
                v = [st[i].permute(1, 2, 0).contiguous() for i in range(4)]
            else:
                v = [st[i].contiguous() for i in range(4)]
            ds['C11'] = (layout, v[0]); ds['C12'] = (layout, torch.complex(v[1], v[2])); ds['C22'] = (layout, v[3])
            algos = [('OmnibusTest a=0.99', OmnibusTest(n=9, alpha=0.99)), ('OmnibusTest a=0.01', OmnibusTest(n=9, alpha=0.01)),
                     ('OmnibusTest ml=3 a=0.01', OmnibusTest(ml=3, alpha=0.01)), ('Boxcar w=3', BoxcarFilter(w=3)),
                     ('Boxcar w=5', BoxcarFilter(w=5)), ('Gaussian s=1', GaussianFilter(sigma=1.0)),
                     ('Convolution rand3x3', ConvolutionFilter(kernel=np.random.default_rng(0).normal(size=(3, 3)))),
                     ('NLMeans tutorial', NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, sigma=0.5, h=0.5, n_eff=50)),
                     ('NLMeans (y,x) r=3', NLMeansFilter(dims=('y', 'x'), r=3, f=1, sigma=0.5, h=0.5))]
            for name, algo in algos:
                ms = t_ms(lambda: algo.apply(ds), 2)
                print('api k=%d layout=%s %-26s: %8.2f ms' % (k, ''.join(d[0] for d in layout), name, ms), flush=True)
            del ds, v
            torch.cuda.empty_cache()
        del st
        torch.cuda.empty_cache()

if what == 'nodata':
    # rasters with nodata margins (NaN or zero fill over 30 % of the pixels, all dates)
    k, ny, nx = 24, 2048, 4096
    base = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=3, device=dev, change_frac=0.01)
    for fill in ('none', 'nan', 'zero', 'nan one date'):
        st = base.clone()
        if fill == 'nan':
            st[:, :, :, : int(0.3 * nx)] = float('nan')
        elif fill == 'zero':
            st[:, :, :, : int(0.3 * nx)] = 0.0
        elif fill == 'nan one date':
            st[:, 5, :, : int(0.3 * nx)] = float('nan')
        for alpha in (0.01, 0.99):
            from nd_amd import _lib
            ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9), 2)
            _lib.timing_enable(64)
            kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
            km = {}
            for n_, v in _lib.timing_collect():
                km[n_] = km.get(n_, 0.0) + v
            _lib.timing_enable(0)
            print('nodata fill=%-13s alpha=%g: %8.2f ms' % (fill, alpha, ms), {a: round(b, 3) for a, b in km.items()}, flush=True)

if what == 'clustered':
    # changes as they occur in real scenes: contiguous regions instead of isolated pixels
    from nd_amd import _lib
    k, ny, nx = 24, 2048, 4096
    base = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=3, device=dev, change_frac=0.0)
    for frac, desc in ((0.0, 'no change'), (0.05, '5 % in one block'), (0.2, '20 % in one block'), (0.5, 'half the raster')):
        st = base.clone()
        rows = int(ny * frac)
        if rows:
            st[:, 12:, 300:300 + rows, :] *= 4.0
        for alpha in (0.01, 0.5, 0.99):
            ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9), 2)
            _lib.timing_enable(64)
            kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
            km = {}
            for n_, v in _lib.timing_collect():
                km[n_] = km.get(n_, 0.0) + v
            _lib.timing_enable(0)
            print('clustered %-18s alpha=%g: %8.2f ms' % (desc, alpha, ms), {a: round(b, 3) for a, b in km.items()}, flush=True)

if what == 'scale':
    # dark targets: small backscatter values, long series (the reference's double product of
    # determinants then leaves the normal range and the fast forms hand the pixel to the exact pass)
    from nd_amd import _lib
    for k in (24, 48, 64):
        ny, nx = 1024, 4096
        base = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=3, device=dev, change_frac=0.01)
        for scale in (1.0, 1e-2, 1e-3, 1e-4):
            st = base * scale
            for alpha in (0.01, 0.99):
                ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9), 1)
                _lib.timing_enable(64)
                ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
                km = {}
                for n_, v in _lib.timing_collect():
                    km[n_] = km.get(n_, 0.0) + v
                _lib.timing_enable(0)
                print('scale k=%d x%g alpha=%g: %8.2f ms changes/px %.2f' % (k, scale, alpha, ms, float(ch.sum()) / (ny * nx)),
                      {a: round(b, 3) for a, b in km.items()}, flush=True)

if what == 'small':
    # interactive sizes: where host overhead, not the kernels, sets the time
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    from nd_amd.filters import BoxcarFilter, NLMeansFilter, GaussianFilter
    import numpy as np
    for ny, nx in ((64, 64), (512, 512), (1024, 1024)):
        k = 24
        st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=12, device=dev, change_frac=0.01)
        for where in ('device', 'host'):
            ds = xr_lite.Dataset()
            v = [st[i].permute(1, 2, 0).contiguous() for i in range(4)]
            c12 = torch.complex(v[1], v[2])
            if where == 'host':
                ds['C11'] = (('y', 'x', 'time'), v[0].cpu().numpy()); ds['C12'] = (('y', 'x', 'time'), c12.cpu().numpy()); ds['C22'] = (('y', 'x', 'time'), v[3].cpu().numpy())
            else:
                ds['C11'] = (('y', 'x', 'time'), v[0]); ds['C12'] = (('y', 'x', 'time'), c12); ds['C22'] = (('y', 'x', 'time'), v[3])
            for name, algo in (('OmnibusTest a=0.01', OmnibusTest(n=9, alpha=0.01)), ('Boxcar w=3', BoxcarFilter(w=3)),
                               ('Gaussian s=1', GaussianFilter(sigma=1.0)),
                               ('NLMeans tutorial', NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, sigma=0.5, h=0.5, n_eff=50))):
                ms = t_ms(lambda: algo.apply(ds), 5)
                print('small %4dx%-4d %-6s %-20s: %8.3f ms' % (ny, nx, where, name, ms), flush=True)

if what == 'nlm_nodata':
    # non-local means on rasters with nodata margins (30 % of the columns NaN / zero on all dates)
    g = torch.Generator(device=dev).manual_seed(5)
    for desc, shape, r, f, sg, h, ne in (('tutorial r=(1,3,3) f=1 V=4', (4, 12, 1024, 4096), (1, 3, 3), (1, 1, 1), 1.0, 1.0, 50.0),
                                         ('7x7 / 21x21 V=1', (1, 6, 2048, 4096), (0, 10, 10), (0, 3, 3), 0.5, 0.5, -1.0)):
        base = torch.rand(shape, generator=g, device=dev) + 0.5
        out = torch.empty_like(base)
        for fill in ('none', 'nan', 'zero'):
            x = base.clone()
            if fill == 'nan':
                x[..., : int(0.3 * shape[-1])] = float('nan')
            elif fill == 'zero':
                x[..., : int(0.3 * shape[-1])] = 0.0
            for pm in (0, 1):
                if r[0] == 0:
                    fn = lambda: kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), out.permute(2, 3, 1, 0), (r[1], r[2], 0), (f[1], f[2], 0), sg, h, ne, patch_mode=pm)
                else:
                    fn = lambda: kernels.pixelwise_nlmeans_3d(x.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, sg, h, ne, patch_mode=pm)
                try:
                    ms = t_ms(fn, 2)
                    print('nlm_nodata %-28s fill=%-5s patch_mode %d: %9.2f ms' % (desc, fill, pm, ms), flush=True)
                except Exception as e:
                    print('nlm_nodata %-28s fill=%-5s patch_mode %d: %s' % (desc, fill, pm, str(e)[:80]), flush=True)
