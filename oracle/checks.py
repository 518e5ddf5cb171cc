"""
oracle/checks.py -- sampled comparisons of device results with the CPU oracle at sizes the oracle
cannot cover whole (one GPU's share of BASELINE.json configs 2-5).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: imported by tests/ and by the checking /
cpu_baseline legs of bench.py (outside every timed region), never by anything under nd_amd/.
Every function takes device tensors, pulls a bounded sample to the host, runs the oracle on it and
returns the number of differing elements (0 = parity) plus what was compared.
"""
import numpy as np
import torch

from . import oracle as O


def _sample_indices(npix, nx, nsample, rows, seed):
    g = torch.Generator(device='cpu').manual_seed(int(seed))
    idx = torch.randint(0, npix, (int(nsample),), generator=g)
    if len(rows):
        r = torch.tensor(list(rows), dtype=torch.int64)
        idx = torch.cat([idx, (r[:, None] * nx + torch.arange(nx)[None]).reshape(-1)])
    return idx


def omnibus_sample(stack, change, alpha, n, nsample=100000, rows=(), seed=5, njobs=8, pol=2):
    """stack: planar (pol*pol, k, ny, nx) device tensor (plane order of the kernels), change: the
    device change map (ny, nx, k).  Compares `nsample` random pixels plus the whole rows `rows`
    with the oracle, byte for byte.  -> dict(bad=, compared=, flagged_fraction=)."""
    nvar, k, ny, nx = stack.shape
    assert nvar == pol * pol
    idx = _sample_indices(ny * nx, nx, nsample, rows, seed)
    dev_idx = idx.to(stack.device)
    yy, xx = dev_idx // nx, dev_idx % nx
    sample = stack[:, :, yy, xx].cpu().numpy()                              # (nvar, k, n)
    planes = [np.ascontiguousarray(sample[v].T)[None] for v in range(nvar)]  # (1, n, k)
    if pol == 2:
        want = O.change_detection_planes(planes, alpha, n, njobs=njobs)[0]
    else:
        want = O.change_detection_pol(planes, pol, alpha, n, njobs=njobs)[0]
    got = change[yy, xx].cpu().numpy()
    res = {'bad': int((got != want).sum()), 'compared': int(got.size),
           'flagged_fraction': float((want.sum(axis=1) > 0).mean())}
    if res['bad']:
        # what went wrong where: enough to tell a stale / unwritten map from a wrong decision
        badpx = np.flatnonzero((got != want).any(axis=1))
        i0 = int(badpx[0])
        res.update(bad_pixels=int(badpx.size), sampled_pixels=int(got.shape[0]),
                   first_bad_pixel=int(idx[i0]), got_first=got[i0].tolist(), want_first=want[i0].tolist(),
                   got_values=np.unique(got)[:8].tolist(),
                   bad_pixel_index_range=[int(idx[badpx].min()), int(idx[badpx].max())])
    return res


def omnibus_ml_bands(stack, change, ml, alpha, bands, njobs=8):
    """OmnibusTest(ml=w) (nd/change.py:61-69): stack planar (4, k, ny, nx) device tensor, change the
    device map (ny, nx, k).  For every (y0, nrows) in `bands` the oracle multilooks the rows
    [y0 - ml // 2, y0 + nrows + ml // 2) of every plane (scipy's boxcar arithmetic; the band keeps real
    neighbour rows, or the raster's own reflection where it touches an edge), tests the band's core
    with n = ml ** 2 and the core's map is compared byte for byte.
    -> dict(bad=, compared=, flagged_fraction=)."""
    nvar, k, ny, nx = stack.shape
    assert nvar == 4
    h = int(ml) // 2
    kern = (np.ones((ml, ml), dtype=np.float64) / ml ** 2).reshape(1, ml, ml)
    bad = comp = flagged = 0
    for y0, nrows in bands:
        a = max(0, min(int(y0), ny - nrows))
        b = a + nrows
        ea, eb = max(a - h, 0), min(b + h, ny)
        sub = stack[:, :, ea:eb, :].cpu().numpy()
        planes = []
        for v in range(4):
            mlv = O.convolve_reflect_mt(np.ascontiguousarray(sub[v]), kern, njobs=njobs)     # (k, rows, nx)
            planes.append(np.ascontiguousarray(np.moveaxis(mlv[:, a - ea:a - ea + nrows, :], 0, -1)))
        with np.errstate(all='ignore'):
            want = O.change_detection_planes(planes, alpha, ml * ml, njobs=njobs)
        got = change[a:b].cpu().numpy()
        bad += int((got != want).sum())
        comp += int(got.size)
        flagged += int((want.sum(axis=2) > 0).sum())
    return {'bad': bad, 'compared': comp, 'flagged_fraction': flagged / max(comp // k, 1)}


def _crop_bounds(lo, n, size, halo):
    """[a, b) = core of a crop, [ea, eb) = core + halo clipped to [0, n)."""
    a = max(0, min(int(lo), n - size))
    b = a + size
    return a, b, max(a - halo, 0), min(b + halo, n)


def nlmeans_crops(stack, filtered, r, f, sigma, h, n_eff, patch_mode, crops, size=(12, 96),
                  njobs=8, then_omnibus=None, change=None):
    """stack / filtered: planar (nvar, k, ny, nx) device tensors, input and output of the device
    filter NLMeansFilter(dims=('time','y','x'), r=(rt,ry,rx), f=(ft,fy,fx)).  For every (y0, x0) in
    `crops` a (size + halo) window is filtered by the oracle (crops touching the raster's edge keep
    the true reflection there) and its core compared with `filtered`: exactly for patch_mode 0,
    to rtol 1e-5 for patch_mode 1.  then_omnibus=(alpha, n): the oracle's change map of the
    oracle-filtered core is also compared with `change` (ny, nx, k), byte for byte.
    -> dict(bad=, compared=, max_rel=, change_bad=, change_compared=)."""
    nvar, k, ny, nx = stack.shape
    rt, ry, rx = (int(v) for v in r)
    ft, fy, fx = (int(v) for v in f)
    hy, hx = ry + fy, rx + fx
    bad = compared = cbad = ccomp = 0
    max_rel = 0.0
    for (y0, x0) in crops:
        a, b, ea, eb = _crop_bounds(y0, ny, size[0], hy)
        c, d, ec, ed = _crop_bounds(x0, nx, size[1], hx)
        win = stack[:, :, ea:eb, ec:ed].cpu().numpy()                      # (nvar, k, Y, X)
        if rt == 0 and ft == 0:
            arr = np.ascontiguousarray(np.transpose(win, (2, 3, 1, 0)))    # (y, x, time, var)
            out = np.empty_like(arr)
            O.pixelwise_nlmeans_3d(arr, out, (ry, rx, 0), (fy, fx, 0), sigma, h, n_eff,
                                   njobs=njobs, patch_mode=patch_mode)
            want = np.transpose(out, (3, 2, 0, 1))
        else:
            arr = np.ascontiguousarray(np.transpose(win, (1, 2, 3, 0)))    # (time, y, x, var)
            out = np.empty_like(arr)
            O.pixelwise_nlmeans_3d(arr, out, (rt, ry, rx), (ft, fy, fx), sigma, h, n_eff,
                                   njobs=njobs, patch_mode=patch_mode)
            want = np.transpose(out, (3, 0, 1, 2))
        core = want[:, :, a - ea:b - ea, c - ec:d - ec]
        got = filtered[:, :, a:b, c:d].cpu().numpy()
        compared += got.size
        if patch_mode == 0:
            bad += int((got != core).sum())
        else:
            # 1e-5 relative -- of the value, or of the magnitude of what was averaged where the average
            # cancels: a filtered cross term (C12 re / im, zero mean) of 1e-4 is a sum of inputs of +-0.3
            # whose float32 weights carry 1e-7 each, so the value is good to ~1e-8 absolute whatever its
            # size.  Floor: 1e-7 of the window's largest magnitude per variable (float32 resolution of the
            # summands); everything above it is held to 1e-5 relative.
            scale = np.abs(win).reshape(win.shape[0], -1).max(axis=1)[:, None, None, None]
            err = np.abs(got - core)
            rel = err / np.maximum(np.abs(core), 1e-30)
            over = (rel > 1e-5) & (err > 1e-7 * scale)
            max_rel = max(max_rel, float(np.where(err > 1e-7 * scale, rel, 0.0).max()))
            bad += int(over.sum())
        if then_omnibus is not None:
            alpha, n = then_omnibus
            planes = [np.ascontiguousarray(np.moveaxis(core[v], 0, -1)) for v in range(4)]
            wantc = O.change_detection_planes(planes, alpha, n, njobs=njobs)
            gotc = change[a:b, c:d].cpu().numpy()
            cbad += int((gotc != wantc).sum())
            ccomp += int(gotc.size)
    res = {'bad': bad, 'compared': compared, 'max_rel': max_rel}
    if then_omnibus is not None:
        res['change_bad'], res['change_compared'] = cbad, ccomp
    return res


def convolve_bands(x, out, kernel2d, bands, dates, halo=12):
    """x / out: (k, ny, nx) device tensors, input and output of a (y, x) convolution with
    `kernel2d`: bands of rows [r0, r1) of a few dates are convolved by the oracle (scipy.ndimage
    arithmetic) with `halo` rows of context and compared exactly.  -> dict(bad=, compared=)."""
    k, ny, nx = x.shape
    bad = compared = 0
    kern = np.asarray(kernel2d, np.float64)
    for t in dates:
        for (r0, r1) in bands:
            e0, e1 = max(r0 - halo, 0), min(r1 + halo, ny)
            host = np.ascontiguousarray(x[t, e0:e1].cpu().numpy())
            want = O.convolve(host, kern)
            lo = halo if e0 > 0 else 0
            hi = want.shape[0] - (halo if e1 < ny else 0)
            got = out[t, e0 + lo:e0 + hi].cpu().numpy()
            bad += int((got != want[lo:hi]).sum())
            compared += int(got.size)
    return {'bad': bad, 'compared': compared}


def gaussian_bands(x, out, sigma, bands, dates, mode='reflect', truncate=4.0):
    """x / out: (k, ny, nx) device tensors, input and output of GaussianFilter(dims=('y','x'),
    sigma): bands of rows [r0, r1) of a few dates are filtered by scipy.ndimage.gaussian_filter
    itself -- the reference's arithmetic for this filter (nd/filters.py:365-378) -- with the
    kernel's reach of extra rows as context, and compared exactly.  -> dict(bad=, compared=)."""
    import scipy.ndimage as ndi
    k, ny, nx = x.shape
    halo = int(truncate * float(sigma) + 0.5) + 1
    bad = compared = 0
    for t in dates:
        for (r0, r1) in bands:
            e0, e1 = max(r0 - halo, 0), min(r1 + halo, ny)
            host = np.ascontiguousarray(x[t, e0:e1].cpu().numpy())
            want = ndi.gaussian_filter(host, sigma, mode=mode, truncate=truncate)
            lo = halo if e0 > 0 else 0
            hi = want.shape[0] - (halo if e1 < ny else 0)
            got = out[t, e0 + lo:e0 + hi].cpu().numpy()
            bad += int((got != want[lo:hi]).sum())
            compared += int(got.size)
    return {'bad': bad, 'compared': compared}
