"""Config C-A restated on the reference's bundled data (last row): a k = 24 series
built from the dual-pol raster the reference ships as data/slc.data (tests/golden/slc_c2: 206 x 500,
73 % exactly-zero nodata margin, backscatter 1e-6 ... 1.4 -- magnitudes the synthetic unit-power
stacks never have), through every layer: the C ABI kernels, OmnibusTest.apply with and without
ml=3, BoxcarFilter and NLMeansFilter, each against the CPU oracle / scipy."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def stack():
    spec = importlib.util.spec_from_file_location(
        'make_slc_stack', os.path.join(HERE, 'golden', 'slc_c2', 'make_slc_stack.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    planes = mod.slc_stack(k=24, looks=9, seed=1)
    assert planes[0].shape == (24, 206, 500) and (planes[0] == 0).mean() > 0.7
    return planes


def _dataset(planes, complex_c12=False):
    from nd_amd import xr_lite
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    ds = xr_lite.Dataset(coords={'y': np.arange(206), 'x': np.arange(500), 'time': np.arange(24)})
    if complex_c12:
        ds['C11'] = (('y', 'x', 'time'), yxt[0])
        ds['C12'] = (('y', 'x', 'time'), (yxt[1] + 1j * yxt[2]).astype(np.complex64))
        ds['C22'] = (('y', 'x', 'time'), yxt[3])
    else:
        for v, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), yxt):
            ds[v] = (('y', 'x', 'time'), a)
    return ds, yxt


@pytest.mark.parametrize('alpha', [1e-4, 0.01, 0.99])
def test_omnibus_on_bundled_raster(stack, oracle, device, alpha):
    import torch
    from nd_amd import kernels
    from nd_amd.change import OmnibusTest
    ds, yxt = _dataset(stack)
    with np.errstate(all='ignore'):
        want, zw, pw = oracle.change_detection_planes(yxt, alpha, 9, njobs=8, stats=True)
    assert want[60:140, 150:350].sum() > 0                      # the injected step is found
    assert want[:, :100].sum() == 0                             # nothing in the nodata margin
    # the plugin surface (pixel-major kernel underneath)
    got = OmnibusTest(n=9, alpha=alpha).apply(ds)
    np.testing.assert_array_equal(got.values, want.astype(bool))
    # complex C12, as the reference's datasets carry it
    got = OmnibusTest(n=9, alpha=alpha).apply(_dataset(stack, complex_c12=True)[0])
    np.testing.assert_array_equal(got.values, want.astype(bool))
    # planar device stack through the C ABI, with the z / P rasters
    dev = [torch.from_numpy(p).to(device) for p in stack]
    ch, z, P = kernels.change_detection(*dev, alpha=alpha, n=9, stats=True)
    np.testing.assert_array_equal(ch.cpu().numpy(), want)
    np.testing.assert_allclose(z.cpu().numpy(), zw, rtol=1e-5, atol=0, equal_nan=True)
    np.testing.assert_allclose(P.cpu().numpy(), pw, rtol=1e-5, atol=1e-7, equal_nan=True)
    ch = kernels.change_detection(*dev, alpha=alpha, n=9)
    np.testing.assert_array_equal(ch.cpu().numpy(), want)


@pytest.mark.parametrize('alpha', [1e-4, 0.01, 0.99])
def test_omnibus_multilooked_bundled_raster(stack, oracle, device, alpha):
    """ml = 3: BoxcarFilter(w=3) then n = 9 (nd/change.py:61-64), on single-look draws."""
    import scipy.ndimage as ndi
    from nd_amd.change import OmnibusTest
    spec = importlib.util.spec_from_file_location(
        'make_slc_stack', os.path.join(HERE, 'golden', 'slc_c2', 'make_slc_stack.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    planes = mod.slc_stack(k=24, looks=1, seed=2)
    ds, yxt = _dataset(planes)
    kern = (np.ones((3, 3)) / 9).reshape(3, 3, 1)
    ml = [ndi.convolve(a, kern) for a in yxt]
    with np.errstate(all='ignore'):
        want = oracle.change_detection_planes(ml, alpha, 9, njobs=8)
    got = OmnibusTest(ml=3, alpha=alpha).apply(ds)
    np.testing.assert_array_equal(got.values, want.astype(bool))
    assert want.sum() > 0
    # the fused multilooking kernel itself, at every threshold (OmnibusTest picks it in the sparse regime)
    import torch
    from nd_amd import kernels
    dev = [torch.from_numpy(np.ascontiguousarray(p)).to(device) for p in planes]
    got = kernels.change_detection_multilooked(*dev, alpha=alpha, ml=3)
    assert got is not None
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_filters_on_bundled_raster(stack, oracle, device):
    import scipy.ndimage as ndi
    from nd_amd.filters import BoxcarFilter, GaussianFilter, NLMeansFilter
    ds, yxt = _dataset(stack)
    got = BoxcarFilter(dims=('y', 'x'), w=5).apply(ds)
    kern = (np.ones((5, 5)) / 25).reshape(5, 5, 1)
    for v, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), yxt):
        np.testing.assert_array_equal(got[v].values, ndi.convolve(a, kern))
    got = GaussianFilter(dims=('y', 'x'), sigma=1.0).apply(ds)
    np.testing.assert_array_equal(got['C22'].values, ndi.gaussian_filter(yxt[3], (1.0, 1.0, 0.0)))
    # non-local means with the tutorial's parameters (examples/tutorial_s1.ipynb cell 11), against
    # the oracle (itself bit-identical to the compiled reference, tests/test_oracle_filters.py)
    crop = {v: a[40:120, 120:260] for v, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), yxt)}
    from nd_amd import xr_lite
    dc = xr_lite.Dataset()
    for v, a in crop.items():
        dc[v] = (('y', 'x', 'time'), np.ascontiguousarray(a))
    for n_eff in (-1, 50):
        got = NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, sigma=1, h=1, n_eff=n_eff).apply(dc)
        arr = np.stack([np.moveaxis(crop[v], -1, 0) for v in ('C11', 'C12__re', 'C12__im', 'C22')], axis=-1)
        arr = np.ascontiguousarray(arr)                          # (time, y, x, variable)
        want = np.empty_like(arr)
        oracle.pixelwise_nlmeans_3d(arr, want, (1, 3, 3), (1, 1, 1), 1.0, 1.0, n_eff, njobs=8)
        for i, v in enumerate(('C11', 'C12__re', 'C12__im', 'C22')):
            np.testing.assert_allclose(np.moveaxis(got[v].values, -1, 0), want[..., i], rtol=1e-5, atol=0)
