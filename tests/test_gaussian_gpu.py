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
