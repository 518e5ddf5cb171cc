"""Randomised parity campaign: the HIP path (through the C ABI) against the CPU oracle on random
shapes, strides, parameters and corner values.  Not part of the pytest suite (run time is open
ended); what it finds is either fixed with a regression case in tests/ or recorded in DESIGN.md 9.

    python tools/fuzz_parity.py --seconds 120 --seed 1 [--what omnibus,nlmeans,correlate,gaussian,c3]

Prints one line per failure (with the parameters needed to replay it) and a summary; exit code 1
if anything differed."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from nd_amd import kernels
from oracle import oracle as O

DEV = torch.device('cuda:0')
ILL_POSED = [0]     # NaN <-> number swaps of the ill-posed n_eff corner (nlmeans)


def wishart(rng, k, ny, nx, looks, dtype, pol=2):
    s = (rng.normal(size=(pol, looks, k, ny, nx)) + 1j * rng.normal(size=(pol, looks, k, ny, nx))) / np.sqrt(2)
    out = {}
    for i in range(pol):
        out['C%d%d' % (i + 1, i + 1)] = (np.abs(s[i]) ** 2).mean(axis=0)
        for j in range(i + 1, pol):
            c = (s[i] * np.conj(s[j])).mean(axis=0)
            out['C%d%dre' % (i + 1, j + 1)] = c.real
            out['C%d%dim' % (i + 1, j + 1)] = c.imag
    return {n: v.astype(dtype) for n, v in out.items()}


def case_omnibus(rng):
    k = int(rng.choice([2, 3, 5, 8, 9, 12, 16, 17, 24, 25, 31, 32, 33, 40, 48, 49, 60, 63, 64, 65, 80, 96, 97, 128, 130, 160, 192]))
    ny, nx = int(rng.integers(1, 40)), int(rng.integers(1, 300))
    looks = int(rng.choice([1, 2, 4, 9, 20]))
    dtype = rng.choice([np.float32, np.float64])
    alpha = float(rng.choice([0.01, 0.05, 0.2, 0.5, 0.7, 0.9, 0.99, 0.999, 0.9999, 1e-4]))
    n = int(rng.choice([looks, 1, 3]))
    w = wishart(rng, k, ny, nx, looks, dtype)
    planes = [w['C11'], w['C12re'], w['C12im'], w['C22']]
    # magnitudes: the fast forms take determinants inside 2^+-36 (float32) and hand the rest to the exact pass
    scale = float(rng.choice([1.0, 1.0, 1.0, 1e-3, 1e-6, 1e4, 1e6]))
    if scale != 1.0:
        planes = [(p * scale).astype(dtype) for p in planes]
    # step changes, zeros, NaNs, infinities, negative determinants
    if rng.random() < 0.7:
        m = rng.random((ny, nx)) < 0.3
        t0 = rng.integers(1, k, (ny, nx)) if k > 1 else np.zeros((ny, nx), int)
        g = np.where((np.arange(k)[:, None, None] >= t0[None]) & m[None], rng.choice([0.1, 4.0, 30.0]), 1.0)
        planes = [(p * g).astype(dtype) for p in planes]
    if rng.random() < 0.3:
        bad = rng.random((k, ny, nx)) < 0.01
        val = rng.choice([0.0, np.nan, np.inf, -1.0])
        planes[int(rng.integers(0, 4))][bad] = val
    layout = rng.choice(['tyx', 'yxt', 'pad', 'pm', 'pmc'])
    desc = dict(k=k, ny=ny, nx=nx, looks=looks, dtype=np.dtype(dtype).name, alpha=alpha, n=n, layout=str(layout))
    with np.errstate(all='ignore'):
        want, zw, pw = O.change_detection_planes([np.moveaxis(p, 0, -1) for p in planes], alpha, n, njobs=8, stats=True)
    if layout in ('tyx', 'pm', 'pmc'):
        dev = [torch.from_numpy(p).to(DEV) for p in planes]
        dims = ('time', 'y', 'x')
    elif layout == 'yxt':
        dev = [torch.from_numpy(np.ascontiguousarray(np.moveaxis(p, 0, -1))).to(DEV) for p in planes]
        dims = ('y', 'x', 'time')
    else:
        big = torch.zeros((4, k, ny + 2, nx + 5), dtype=torch.from_numpy(planes[0]).dtype, device=DEV)
        for v in range(4):
            big[v, :, 1:ny + 1, 3:nx + 3] = torch.from_numpy(planes[v]).to(DEV)
        dev = [big[v, :, 1:ny + 1, 3:nx + 3] for v in range(4)]
        dims = ('time', 'y', 'x')
    stats = bool(rng.random() < 0.5)
    ws = str(rng.choice(['recommended', 'minimal']))
    desc.update(stats=stats, workspace=ws)
    res = None
    if layout in ('pm', 'pmc'):
        # the reference's own layout through the pixel-major kernel (C12 interleaved for 'pmc')
        dev = [torch.from_numpy(np.ascontiguousarray(np.moveaxis(p, 0, -1))).to(DEV) for p in planes]
        dims = ('y', 'x', 'time')
        if layout == 'pmc':
            c12 = torch.complex(dev[1], dev[2])
            dev = [dev[0], c12.real, c12.imag, dev[3]]
        res = kernels.change_detection_pixel_major(*dev, alpha=alpha, n=n, stats=stats)
        if res is None and layout == 'pmc':
            dev = [d.contiguous() for d in dev]
    if res is None:
        res = kernels.change_detection(*dev, alpha=alpha, n=n, dims=dims, stats=stats, workspace=ws)
    got = (res[0] if stats else res).cpu().numpy()
    ok = np.array_equal(got, want)
    if ok and stats:
        z, P = res[1].cpu().numpy(), res[2].cpu().numpy()
        with np.errstate(all='ignore'):
            ok = (np.allclose(z, zw, rtol=1e-5, atol=0, equal_nan=True) and
                  np.allclose(P, pw, rtol=1e-5, atol=1e-7, equal_nan=True))
    return ok, desc


def case_omnibus_ml(rng):
    """OmnibusTest(ml=w): the fused multilooking kernel (nd_amd_omnibus_c2_ml) against scipy's boxcar
    followed by the oracle's test with n = ml ** 2 (nd/change.py:61-69)."""
    import scipy.ndimage as ndi
    ml = int(rng.choice([3, 5]))
    k = int(rng.choice([3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 23, 24]))
    ny, nx = int(rng.integers(ml, 60)), int(rng.integers(ml, 400))
    if rng.random() < 0.3:
        nx = int(rng.choice([64, 128, 192, 256, 68, 132]))        # whole tiles, 16-byte rows
    looks = int(rng.choice([1, 1, 2, 4]))
    alpha = float(rng.choice([0.01, 0.2, 0.7, 0.9, 0.95, 0.99, 0.999, 1e-4]))
    w = wishart(rng, k, ny, nx, looks, np.float32)
    planes = [w['C11'], w['C12re'], w['C12im'], w['C22']]
    scale = float(rng.choice([1.0, 1.0, 1.0, 1e-3, 1e-6, 1e4]))
    if scale != 1.0:
        planes = [(p * np.float32(scale)).astype(np.float32) for p in planes]
    if rng.random() < 0.7:
        m = rng.random((ny, nx)) < 0.3
        t0 = rng.integers(1, k, (ny, nx))
        g = np.where((np.arange(k)[:, None, None] >= t0[None]) & m[None], rng.choice([0.1, 4.0, 30.0]), 1.0)
        planes = [(p * g).astype(np.float32) for p in planes]
    if rng.random() < 0.3:
        bad = rng.random((k, ny, nx)) < 0.005
        planes[int(rng.integers(0, 4))][bad] = rng.choice([0.0, np.nan, np.inf, -1.0])
    if rng.random() < 0.2:                                          # a nodata margin
        for p in planes:
            p[:, :, : nx // 3] = 0
    stats = bool(rng.random() < 0.4)
    layout = str(rng.choice(['tyx', 'pad']))
    desc = dict(ml=ml, k=k, ny=ny, nx=nx, looks=looks, alpha=alpha, stats=stats, layout=layout, scale=scale)
    kern = (np.ones((ml, ml), dtype=np.float64) / ml ** 2).reshape(1, ml, ml)
    with np.errstate(all='ignore'):
        mlp = [ndi.convolve(p, kern) for p in planes]
        want, zw, pw = O.change_detection_planes([np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in mlp],
                                                 alpha, ml * ml, njobs=8, stats=True)
    if layout == 'tyx':
        dev = [torch.from_numpy(p).to(DEV) for p in planes]
    else:
        big = torch.zeros((4, k, ny + 2, nx + 8), dtype=torch.float32, device=DEV)
        for v in range(4):
            big[v, :, 1:ny + 1, 4:nx + 4] = torch.from_numpy(planes[v]).to(DEV)
        dev = [big[v, :, 1:ny + 1, 4:nx + 4] for v in range(4)]
    res = kernels.change_detection_multilooked(*dev, alpha=alpha, ml=ml, stats=stats)
    if res is None:
        return False, dict(desc, refused=True)
    got = (res[0] if stats else res).cpu().numpy()
    ok = np.array_equal(got, want)
    if ok and stats:
        z, P = res[1].cpu().numpy(), res[2].cpu().numpy()
        with np.errstate(all='ignore'):
            ok = (np.allclose(z, zw, rtol=1e-5, atol=0, equal_nan=True) and
                  np.allclose(P, pw, rtol=1e-5, atol=1e-7, equal_nan=True))
    return ok, desc


def case_c3(rng):
    k = int(rng.choice([2, 3, 4, 7, 12, 24, 33, 48, 63, 64, 65]))
    ny, nx = int(rng.integers(1, 20)), int(rng.integers(1, 200))
    looks = int(rng.choice([3, 9, 16]))
    dtype = rng.choice([np.float32, np.float64])
    alpha = float(rng.choice([1e-4, 0.01, 0.1, 0.5, 0.9, 0.99, 0.9999]))
    w = wishart(rng, k, ny, nx, looks, dtype, pol=3)
    names = ['C11', 'C22', 'C33', 'C12re', 'C12im', 'C13re', 'C13im', 'C23re', 'C23im']
    planes = [w[n] for n in names]
    if rng.random() < 0.7:
        m = rng.random((ny, nx)) < 0.3
        t0 = rng.integers(1, k, (ny, nx))
        g = np.where((np.arange(k)[:, None, None] >= t0[None]) & m[None], 5.0, 1.0)
        planes = [(p * g).astype(dtype) for p in planes]
    if rng.random() < 0.3:       # zeros, NaNs, infinities, negative entries, a global scale
        bad = rng.random((k, ny, nx)) < 0.01
        planes[int(rng.integers(0, 9))][bad] = rng.choice([0.0, np.nan, np.inf, -1.0])
        planes = [(p * rng.choice([1.0, 1e-6, 1e5])).astype(dtype) for p in planes]
    desc = dict(k=k, ny=ny, nx=nx, looks=looks, dtype=np.dtype(dtype).name, alpha=alpha)
    with np.errstate(all='ignore'):
        want = O.change_detection_pol([np.moveaxis(p, 0, -1) for p in planes], 3, alpha, looks, njobs=8)
    dev = [torch.from_numpy(p).to(DEV) for p in planes]
    got = kernels.change_detection_c3(dev, alpha=alpha, n=looks).cpu().numpy()
    return np.array_equal(got, want), desc


def case_nlmeans(rng):
    layout = str(rng.choice(['A', 'B', 'C']))
    nv = int(rng.choice([1, 1, 2, 4, 5]))
    pm = int(rng.choice([0, 1]))
    dtype = np.float32 if rng.random() < 0.8 else np.float64
    if layout == 'A':            # (y, x, time, var) view of planar memory, 2-D window
        shape = (int(rng.integers(12, 80)), int(rng.integers(12, 200)), int(rng.integers(1, 4)), nv)
        r = (int(rng.integers(0, 6)), int(rng.integers(0, 6)), 0)
        f = (int(rng.integers(0, 3)),) * 2 + (0,) if rng.random() < 0.7 else (int(rng.integers(0, 3)), int(rng.integers(0, 3)), 0)
        perm = (3, 2, 0, 1)
    elif layout == 'B':          # (time, y, x, var) view of planar memory, 3-D window
        shape = (int(rng.integers(3, 8)), int(rng.integers(12, 60)), int(rng.integers(12, 200)), nv)
        r = (int(rng.integers(0, 3)), int(rng.integers(0, 5)), int(rng.integers(0, 5)))
        f = tuple(int(v) for v in rng.integers(0, 2, 3))
        if rng.random() < 0.35:  # three-date square windows: the streaming window kernel (patch_mode 0)
            R = int(rng.integers(1, 6))
            shape = (int(rng.integers(2, 9)), int(rng.integers(2 * R + 2, 70)), int(rng.integers(2 * R + 2, 300)), nv)
            r, f, pm = (1, R, R), (int(rng.integers(0, 2)), 1, int(rng.integers(1, 3))), 0
        perm = (3, 0, 1, 2)
    else:                        # C-contiguous (a, b, c, var): generic kernel
        shape = (int(rng.integers(3, 12)), int(rng.integers(5, 30)), int(rng.integers(5, 40)), nv)
        r = tuple(int(v) for v in rng.integers(0, 3, 3))
        f = tuple(int(v) for v in rng.integers(0, 2, 3))
        perm = None
    for d in range(3):           # keep a single reflection inside the array
        reach = r[d] + f[d]
        if reach > shape[d] - 1:
            r = tuple(0 if i == d else r[i] for i in range(3)); f = tuple(0 if i == d else f[i] for i in range(3))
    sigma, h = float(rng.choice([0.3, 0.5, 1.0, 2.0])), float(rng.choice([0.3, 0.5, 1.0, 2.0]))
    ne = float(rng.choice([-1, -1, 3.0, 10.0, 50.0, 1.0]))
    nq = (2 * r[0] + 1) * (2 * r[1] + 1) * (2 * r[2] + 1) - 1
    if ne == nq + 1:
        # ill-posed corner of find_weight (nd/_filters.pyx:296-315): with n_eff - 1 equal to the
        # number of neighbours, equal weights put the discriminant at exactly 0 +- rounding noise,
        # and whether the reference returns a number or NaN depends on the last ulp of libm's exp
        ne = float(nq + 2)
    a = rng.gamma(4.0, 0.25, shape).astype(dtype)
    if rng.random() < 0.3:
        a -= a.mean().astype(dtype)
    nodata = 'none'
    if rng.random() < 0.3:       # nodata: a NaN margin, isolated NaNs, infinities of either sign (inf - inf)
        nodata = str(rng.choice(['margin', 'points', 'inf', 'mixed']))
        if nodata in ('margin', 'mixed'):
            ax = int(rng.integers(0, 3))
            sl = [slice(None)] * 4
            sl[ax] = slice(0, max(1, shape[ax] // 4))
            a[tuple(sl)] = np.nan
        if nodata in ('points', 'mixed'):
            a[rng.random(shape) < 0.003] = np.nan
        if nodata in ('inf', 'mixed'):
            m = rng.random(shape) < 0.004
            a[m] = np.where(rng.random(int(m.sum())) < 0.7, np.inf, -np.inf).astype(dtype)
    desc = dict(layout=layout, shape=shape, r=r, f=f, sigma=sigma, h=h, n_eff=ne, patch_mode=pm, dtype=np.dtype(dtype).name,
                nodata=nodata)
    want = np.empty_like(a)
    with np.errstate(all='ignore'):
        O.pixelwise_nlmeans_3d(a, want, r, f, sigma, h, ne, neff_policy=0, njobs=8, patch_mode=pm)
    t = torch.from_numpy(a).to(DEV)
    if perm is not None:
        inv = [perm.index(i) for i in range(4)]
        t = t.permute(*perm).contiguous().permute(*inv)
    out = torch.empty_like(t)
    kernels.pixelwise_nlmeans_3d(t, out, r, f, sigma, h, ne, patch_mode=pm, neff_policy=0)
    got = out.cpu().numpy()
    uniform = pm == 0 and max(f) > 0
    if uniform or dtype == np.float64 and False:
        ok = np.array_equal(got, want, equal_nan=True)
    else:
        fin = a[np.isfinite(a)]
        scale = float(np.abs(fin).max()) if fin.size else 1.0
        with np.errstate(all='ignore'):
            close = np.isclose(got, want, rtol=1e-5, atol=2e-6 * scale, equal_nan=True)
        ok = bool(close.all())
        if not ok and ne >= 0 and nodata != 'none':
            # Infinite data give weights of exactly 0, so the number of neighbours that carry weight
            # can equal n_eff - 1 anywhere (the generator only avoids that for the whole window), e.g.
            # a sample and its own reflection at the raster's edge: W^2 / W2 = n_eff - 1 +- rounding
            # noise decides between find_weight's error branch and a number.  The pixels where the
            # ORACLE ITSELF changes when n_eff moves by 1e-9 are that corner, not comparable.
            lo, hi = np.empty_like(a), np.empty_like(a)
            with np.errstate(all='ignore'):
                O.pixelwise_nlmeans_3d(a, lo, r, f, sigma, h, ne * (1 - 1e-9), neff_policy=0, njobs=8, patch_mode=pm)
                O.pixelwise_nlmeans_3d(a, hi, r, f, sigma, h, ne * (1 + 1e-9), neff_policy=0, njobs=8, patch_mode=pm)
                unstable = ~np.isclose(lo, hi, rtol=1e-6, atol=2e-7 * scale, equal_nan=True)
            if (close | unstable).all() and unstable.sum() <= max(16, got.size // 5):
                ILL_POSED[0] += int((unstable & ~close).sum())
                ok = True
        if not ok and ne >= 0:
            # find_weight's discriminant n tw^2 - n (n - 1) tsq is exactly 0 +- rounding noise when
            # n_eff - 1 neighbours share all the weight (e.g. a sample and its own reflection at the
            # raster's edge): NaN or a number, decided by the last ulp of libm's exp in the
            # reference itself.  Such NaN <-> number swaps are counted, not failed.
            swaps = np.isnan(got) != np.isnan(want)
            if (close | swaps).all() and swaps.sum() <= max(4, got.size // 200):
                ILL_POSED[0] += int(swaps.sum())
                ok = True
    return ok, desc


def case_correlate(rng):
    nd = int(rng.choice([2, 3]))
    dtype = np.float32 if rng.random() < 0.7 else np.float64
    shape = tuple(int(v) for v in (rng.integers(1, 6, nd - 2).tolist() + [rng.integers(1, 70), rng.integers(1, 300)]))
    ksh = tuple(int(v) for v in ([1] * (nd - 2) if rng.random() < 0.7 else rng.integers(1, 4, nd - 2).tolist()) +
                rng.integers(1, 8, 2).tolist())
    kind = str(rng.choice(['box', 'rand', 'sparse']))
    origin = 0
    if rng.random() < 0.35:      # square 3 / 5 / 7 windows on float32 planes: the register-window kernel
        w = int(rng.choice([3, 5, 7]))
        dtype = np.float32
        shape = shape[:-2] + (int(rng.integers(1, 90)), int(rng.integers(8, 900)))
        ksh = (1,) * (nd - 2) + (w, w)
        kind = str(rng.choice(['box', 'rand']))
        if rng.random() < 0.3:
            origin = (0,) * (nd - 2) + (int(rng.integers(-(w // 2), w // 2 + 1)), 0)
    if kind == 'box':
        kern = np.ones(ksh) / np.prod(ksh)
    else:
        kern = rng.normal(size=ksh)
        if kind == 'sparse':
            kern[rng.random(ksh) < 0.4] = 0.0
            if not kern.any():
                kern.flat[0] = 1.0
    mode = str(rng.choice(['reflect', 'constant', 'nearest', 'mirror', 'wrap']))
    cval = float(rng.choice([0.0, 1.5]))
    a = rng.normal(size=shape).astype(dtype)
    desc = dict(shape=shape, kernel=ksh, kind=kind, mode=mode, cval=cval, origin=origin, dtype=np.dtype(dtype).name)
    want = O.convolve(a, kern, mode=mode, cval=cval, origin=origin)
    t = torch.from_numpy(a).to(DEV)
    if rng.random() < 0.3 and nd == 3:          # (y, x, time)-style memory: window axes are not the fastest
        t = t.permute(1, 2, 0).contiguous().permute(2, 0, 1)
        desc['strided'] = True
    got = kernels.convolve(t, kern, mode=mode, cval=cval, origin=origin).cpu().numpy()
    return np.array_equal(got, want, equal_nan=True), desc


def case_gaussian(rng):
    import scipy.ndimage as snf
    nd = int(rng.choice([2, 3]))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    shape = tuple(int(v) for v in rng.integers(1, 60, nd))
    sigma = tuple(float(v) for v in rng.choice([0.0, 0.5, 1.0, 2.5, 7.0], nd))
    mode = str(rng.choice(['reflect', 'constant', 'nearest', 'mirror', 'wrap']))
    fused = rng.random() < 0.45  # (y, x) passes of one radius on float32 planes: the fused kernel
    if fused:
        dtype = np.float32
        sg = float(rng.choice([0.3, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0]))
        shape = (int(rng.integers(1, 4)), int(rng.integers(1, 90)), int(rng.integers(8, 800)))
        sigma = (0.0, sg, sg if rng.random() < 0.8 else sg * 1.05)
        mode = str(rng.choice(['reflect', 'nearest', 'mirror', 'wrap']))
    a = rng.normal(size=shape).astype(dtype)
    if fused and rng.random() < 0.3:
        a[rng.random(shape) < 0.002] = rng.choice([np.inf, -np.inf, np.nan, 0.0])
    desc = dict(shape=shape, sigma=sigma, mode=mode, dtype=np.dtype(dtype).name)
    with np.errstate(all='ignore'):
        want = snf.gaussian_filter(a, sigma=sigma, mode=mode)
    got = kernels.gaussian_filter(torch.from_numpy(a).to(DEV), sigma, mode=mode).cpu().numpy()
    return np.array_equal(got, want, equal_nan=True), desc


CASES = {'omnibus': case_omnibus, 'omnibus_ml': case_omnibus_ml, 'c3': case_c3, 'nlmeans': case_nlmeans, 'correlate': case_correlate,
         'gaussian': case_gaussian}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=60.0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--what', default=','.join(CASES))
    a = ap.parse_args()
    names = a.what.split(',')
    O.build()
    count = {n: 0 for n in names}
    fails = 0
    t_end = time.time() + a.seconds
    i = 0
    t_say = time.time() + 60.0
    while time.time() < t_end:
        if time.time() > t_say:              # a heartbeat: silent runs are taken for hung ones
            print('progress', count, 'failures', fails, flush=True)
            t_say = time.time() + 60.0
        name = names[i % len(names)]
        rng = np.random.default_rng([a.seed, i])
        try:
            ok, desc = CASES[name](rng)
        except Exception as e:                       # an error is a failure too
            ok, desc = False, {'exception': repr(e)}
        count[name] += 1
        if not ok:
            fails += 1
            print('FAIL %s seed=(%d,%d) %s' % (name, a.seed, i, desc), flush=True)
        i += 1
    print('cases', count, 'failures', fails, 'ill-posed n_eff swaps (not failures)', ILL_POSED[0])
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
