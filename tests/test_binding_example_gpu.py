"""The reference-side binding of INTEGRATION.md, executed: every function of examples/nd_binding.py
(numpy in, numpy out, exactly the arguments the reference passes at its four native call sites)
against the oracle, the golden outputs of the real compiled reference, and scipy itself."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'nlmeans_ref.npz')


@pytest.fixture(scope='module')
def binding(device):
    spec = importlib.util.spec_from_file_location('nd_binding_example',
                                                  os.path.join(ROOT, 'examples', 'nd_binding.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_change_detection_binding(binding, oracle):
    """nd/change.py:69: (y, x, time) variables, C12 complex64, as nd/change.py:59-67 holds them."""
    from tests import synth
    for k, alpha in ((12, 0.9), (24, 0.99), (8, 0.01)):
        planes = synth.omnibus_stack(seed=5 + k, k=k, ny=37, nx=53, dtype=np.float32, change_frac=0.15)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        c12 = (yxt[1] + 1j * yxt[2]).astype(np.complex64)
        got = binding.change_detection(yxt[0], c12, yxt[3], alpha, n=9)
        want = oracle.change_detection_planes(yxt, alpha, 9)
        assert got.dtype == np.uint8 and got.shape == (37, 53, k)
        np.testing.assert_array_equal(got, want)
        assert want.sum() > 0


def test_change_detection_multilooked_binding(binding, oracle):
    """nd/change.py:61-69 with ml: scipy's boxcar on every plane, n = ml ** 2, the test."""
    import scipy.ndimage as ndi
    from tests import synth
    for k, alpha, ml in ((12, 0.9, 3), (24, 0.99, 5), (8, 0.01, 3)):
        planes = synth.omnibus_stack(seed=9 + k, k=k, ny=41, nx=150, looks=1, dtype=np.float32, change_frac=0.15)
        got = binding.change_detection_multilooked(*planes, alpha, ml)
        kern = (np.ones((ml, ml)) / ml ** 2).reshape(1, ml, ml)
        mlp = [np.ascontiguousarray(np.moveaxis(ndi.convolve(p, kern), 0, -1)) for p in planes]
        want = oracle.change_detection_planes(mlp, alpha, ml * ml)
        assert got is not None and got.dtype == np.uint8 and got.shape == (41, 150, k)
        np.testing.assert_array_equal(got, want)
    assert binding.change_detection_multilooked(*planes, 0.9, 4) is None          # even window: not covered


def test_convolve_binding(binding):
    """nd/filters.py:262-267: scipy.ndimage.convolve with the kernel broadcast to arr.ndim."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(3)
    arr = rng.gamma(4.0, 0.25, (31, 45, 6)).astype(np.float32)
    cases = [(np.ones((5, 5, 1)) / 25.0, {}),                              # BoxcarFilter(w=5)
             (rng.normal(size=(3, 3, 1)), {}),                             # ConvolutionFilter
             (rng.normal(size=(4, 2, 1)), {}),                             # even sizes: origin shift
             (rng.normal(size=(3, 5, 1)), {'mode': 'nearest'}),
             (rng.normal(size=(3, 3, 1)), {'mode': 'constant', 'cval': 1.5})]
    for kern, kw in cases:
        want = ndi.convolve(arr, kern, **kw)
        got = np.empty_like(arr)
        binding.convolve(arr, kern, got, **kw)
        np.testing.assert_array_equal(got, want)
    a64 = rng.normal(size=(20, 21))
    got = np.empty_like(a64)
    binding.convolve(a64, np.ones((3, 3)) / 9.0, got)
    np.testing.assert_array_equal(got, ndi.convolve(a64, np.ones((3, 3)) / 9.0))


def test_nlmeans_binding(binding):
    """nd/filters.py:462 against the outputs of the real compiled reference (oracle/_ref)."""
    g = np.load(GOLD)
    for name in sorted({n.split('__')[0] for n in g.files if '__' in n}):
        a, par, want = g[name + '__in'], g[name + '__par'], g[name + '__out']
        out = np.empty_like(a)
        binding._pixelwise_nlmeans_3d(a, out, par[:3].astype(np.uint32), par[3:6].astype(np.uint32),
                                      float(par[6]), float(par[7]), float(par[8]))
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=0)
    # find_weight without a solution: ValueError('No solution'), as a build of the reference raises
    a = np.random.default_rng(0).gamma(4.0, 0.25, (8, 8, 1, 1))
    with pytest.raises(ValueError, match='No solution'):
        binding._pixelwise_nlmeans_3d(a, np.empty_like(a), (1, 1, 0), (0, 0, 0), 1.0, 1.0, 50.0)


def test_gaussian_binding(binding):
    """nd/filters.py:372-378: bit-equal to scipy.ndimage.gaussian_filter."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(4)
    arr = rng.gamma(4.0, 0.25, (33, 40, 5)).astype(np.float32)
    for sigma, kw in (((1.0, 1.0, 0.0), {}), ((0.6, 2.0, 0.0), {}), ((1.5, 0.0, 0.8), {'mode': 'nearest'})):
        want = ndi.gaussian_filter(arr, sigma=sigma, **kw)
        got = np.empty_like(arr)
        binding.gaussian_filter(arr, sigma, got, **kw)
        np.testing.assert_array_equal(got, want)
    a64 = rng.normal(size=(25, 30))
    got = np.empty_like(a64)
    binding.gaussian_filter(a64, 1.2, got)
    np.testing.assert_array_equal(got, ndi.gaussian_filter(a64, 1.2))
