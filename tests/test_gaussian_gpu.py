"""GaussianFilter on the GPU: bit-equal to scipy.ndimage.gaussian_filter
(nd/tests/test_gaussian_filter.py:10-27 demand equality with scipy)."""
from collections import OrderedDict

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _gpu_gauss(a, sigma, device, **kw):
    import torch
    from nd_amd import kernels
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    out = kernels.gaussian_filter(t, sigma, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_against_scipy(device, dtype):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(2)
    for shape, sigma in [((20, 20, 10), (1, 1, 0)), ((33, 17), 2.5), ((5, 40, 41), (0, 1.5, 0.7)),
                         ((64,), 3), ((3, 4, 5, 6), (0, 1, 0, 2)), ((10, 10), (0, 0)), ((4, 6), 8.0)]:
        a = rng.normal(size=shape).astype(dtype)
        np.testing.assert_array_equal(_gpu_gauss(a, sigma, device), ndi.gaussian_filter(a, sigma))
    a = rng.normal(size=(30, 31)).astype(dtype)
    for mode in ('reflect', 'constant', 'nearest', 'mirror', 'wrap'):
        np.testing.assert_array_equal(_gpu_gauss(a, 1.3, device, mode=mode, cval=0.3, truncate=3.0),
                                      ndi.gaussian_filter(a, 1.3, mode=mode, cval=0.3, truncate=3.0))


def test_correlate1d_general_kernels(device):
    import torch
    import scipy.ndimage as ndi
    from nd_amd import kernels
    rng = np.random.default_rng(3)
    a = rng.normal(size=(13, 29)).astype(np.float64)
    t = torch.from_numpy(a).to(device)
    for w in (rng.normal(size=5), rng.normal(size=4), np.array([1.0, 0.0, -1.0]), np.array([0.25, 0.5, 0.25])):
        for axis in (0, 1):
            out = torch.empty_like(t)
            kernels.correlate1d(t, w, axis, out)
            np.testing.assert_array_equal(out.cpu().numpy(), ndi.correlate1d(a, w, axis=axis))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_long_kernels_in_the_tiled_1d_kernel(device, dtype):
    """Kernels of 33 .. 97 taps (round 6: Gaussian sigma > 3.75 took the per-element kernel): sigma 5 / 8 / 12 along
    every axis of 3-D arrays (size classes 65 and 97, the filtered axis shorter than the kernel included), general
    (non-symmetric, antisymmetric, even) kernels of up to 97 taps, the border modes -- bit-equal to scipy."""
    import time
    import torch
    import scipy.ndimage as ndi
    from nd_amd import kernels
    rng = np.random.default_rng(97)
    for shape in [(3, 150, 200), (2, 40, 300), (20, 33, 129)]:
        a = rng.normal(size=shape).astype(dtype)
        for sigma in ((0, 5, 5), (0, 8, 0), (0, 0, 12), (5, 0, 0), (3.9, 4.1, 6.0)):
            for mode in ('reflect', 'wrap', 'constant'):
                np.testing.assert_array_equal(_gpu_gauss(a, sigma, device, mode=mode, cval=0.5),
                                              ndi.gaussian_filter(a, sigma, mode=mode, cval=0.5), err_msg=str((shape, sigma, mode)))
    a = rng.normal(size=(60, 170)).astype(dtype)
    t = torch.from_numpy(a).to(device)
    anti = rng.normal(size=48)
    anti = np.concatenate([-anti[::-1], [0.0], anti])
    for w in (rng.normal(size=33), rng.normal(size=64), rng.normal(size=97), anti, np.ones(41) / 41):
        for axis in (0, 1):
            for mode in ('nearest', 'mirror'):
                out = torch.empty_like(t)
                kernels.correlate1d(t, w, axis, out, mode)
                np.testing.assert_array_equal(out.cpu().numpy(), ndi.correlate1d(a, w, axis=axis, mode=mode))
    # the tiled kernel took them: sigma = 8 on 4 x 1024^2 in well under 5 ms
    x = torch.from_numpy(rng.normal(size=(4, 1024, 1024)).astype(dtype)).to(device)
    kernels.gaussian_filter(x, (0, 8, 8))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernels.gaussian_filter(x, (0, 8, 8))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5e-3


def test_gaussian_filter_class(device):
    """nd/tests/test_gaussian_filter.py:10-27 re-stated."""
    import scipy.ndimage as ndi
    from nd_amd.filters import GaussianFilter, gaussian
    ds = synth.lite_test_dataset(dims=OrderedDict([('y', 20), ('x', 20), ('time', 10)]))
    out = GaussianFilter(dims=('y', 'x'), sigma=1).apply(ds)
    np.testing.assert_array_equal(out.C11.values, ndi.gaussian_filter(ds.C11.values, (1, 1, 0)))
    out2 = gaussian(ds, dims=('y', 'x', 'time'), sigma=(1, 2, 0.5))
    np.testing.assert_array_equal(out2.C22.values, ndi.gaussian_filter(ds.C22.values, (1, 2, 0.5)))
    # njobs: halo int(4 sigma + 0.5) per chunk
    a = GaussianFilter(dims=('y', 'x'), sigma=1).apply(ds, njobs=2)
    for v in ds.data_vars:
        np.testing.assert_allclose(a[v].values, out[v].values, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('npdt', [np.float32, np.float64])
@pytest.mark.parametrize('mode', ['reflect', 'nearest', 'mirror', 'wrap'])
def test_fused_y_then_x_kernel(device, mode, npdt):
    """GaussianFilter(dims=('y', 'x')) on x-contiguous float32 / float64 planes runs both passes in one
    kernel (nd_amd_correlate1d_yx); every radius it is instantiated for, strip borders (248 / 240
    written columns per wave), planes shorter than the kernel, unaligned pitches, non-finite
    values -- bit-equal to scipy, whose intermediate array is in the array dtype too."""
    import scipy.ndimage as ndi
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(41)
    tdt = torch.float32 if npdt == np.float32 else torch.float64
    for shape in [(2, 70, 248), (1, 40, 249), (3, 5, 500), (1, 3, 9), (2, 140, 8), (1, 33, 1003), (2, 9, 241)]:
        a = rng.normal(size=shape).astype(npdt)
        for sigma in (0.3, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0):         # radii 1, 2, 3, 4, 5, 6, 8
            np.testing.assert_array_equal(_gpu_gauss(a, (0, sigma, sigma), device, mode=mode),
                                          ndi.gaussian_filter(a, (0, sigma, sigma), mode=mode))
    a = rng.normal(size=(2, 60, 300)).astype(npdt)
    # different sigmas with one radius (fused), with different radii (two passes), truncate
    for sigma, tr in (((0, 1.0, 1.1), 4.0), ((0, 1.0, 2.0), 4.0), ((0, 1.6, 1.6), 2.0)):
        np.testing.assert_array_equal(_gpu_gauss(a, sigma, device, mode=mode, truncate=tr),
                                      ndi.gaussian_filter(a, sigma, mode=mode, truncate=tr))
    b = a.copy()
    b[0, 10, 20] = np.inf
    b[1, 30, 200] = np.nan
    b[1, 5, 5] = -np.inf
    with np.errstate(all='ignore'):
        want = ndi.gaussian_filter(b, (0, 1, 1), mode=mode)
    got = _gpu_gauss(b, (0, 1, 1), device, mode=mode)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    # views with odd pitches in and out
    base = torch.from_numpy(rng.normal(size=(2, 50, 263)).astype(npdt)).to(device)
    view = base[:, 3:47, 1:260]
    obuf = torch.empty((2, 44, 261), dtype=tdt, device=device)
    out = obuf[:, :, :259]
    kernels.gaussian_filter(view, (0, 1, 1), out=out, mode=mode)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(),
                                  ndi.gaussian_filter(view.cpu().numpy(), (0, 1, 1), mode=mode))
