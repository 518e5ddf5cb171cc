"""Pins of the nlmeans and convolution oracles (CPU): the real reference (golden vectors from
oracle/_ref, and oracle/_ref itself when it is present) and scipy.ndimage.convolve."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GN = os.path.join(os.path.dirname(__file__), 'golden', 'nlmeans_ref.npz')
GC = os.path.join(os.path.dirname(__file__), 'golden', 'convolve_scipy.npz')


def _nlm_cases():
    g = np.load(GN)
    return sorted({n.split('__')[0] for n in g.files if '__' in n})


@pytest.mark.parametrize('name', _nlm_cases())
def test_nlmeans_golden_bit_exact(oracle, name):
    g = np.load(GN)
    a, par, want = g[name + '__in'], g[name + '__par'], g[name + '__out']
    out = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, out, par[:3].astype(int), par[3:6].astype(int), par[6], par[7],
                                par[8], njobs=4)
    np.testing.assert_array_equal(out, want)


def test_nlmeans_patch_mode_1_pinned_through_reference(oracle):
    g = np.load(GN)
    a, par, want = g['pm1_in'], g['pm1_par'], g['pm1_out_interior']
    r, f = par[:3].astype(int), par[3:6].astype(int)
    arr = np.ascontiguousarray(a[:, :, None, None])
    out = np.empty_like(arr)
    oracle.pixelwise_nlmeans_3d(arr, out, r, f, par[6], par[7], par[8], patch_mode=1)
    m, n = r[0] + f[0], r[1] + f[1]
    np.testing.assert_array_equal(out[m:-m, n:-n, 0, 0], want)
    # and the compiled-reference semantics differ from it here (the patch loops are empty)
    out0 = np.empty_like(arr)
    oracle.pixelwise_nlmeans_3d(arr, out0, r, f, par[6], par[7], par[8], patch_mode=0)
    assert not np.array_equal(out0, out)


@pytest.mark.skipif(not glob.glob(os.path.join(ROOT, 'oracle', '_ref', 'nd', '_filters*.so')),
                    reason='oracle/_ref not built (needs /root/reference)')
def test_nlmeans_live_reference(oracle):
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    from nd import _filters as RF
    rng = np.random.default_rng(31)
    for dt in (np.float32, np.float64):
        a = rng.normal(0, 1, (9, 11, 4, 2)).astype(dt)
        for r, f, s, h, ne in [((2, 1, 1), (1, 1, 1), 0.5, 0.7, -1), ((2, 2, 1), (0, 0, 0), 0.5, 0.7, -1),
                               ((1, 2, 0), (0, 0, 0), 1.0, 2.0, 3.0)]:
            want = np.empty_like(a)
            RF._pixelwise_nlmeans_3d(a, want, np.array(r, np.uint32), np.array(f, np.uint32), s, h, ne)
            got = np.empty_like(a)
            oracle.pixelwise_nlmeans_3d(a, got, r, f, s, h, ne)
            np.testing.assert_array_equal(got, want)
    assert RF.find_weight(10, 5, 3) == oracle.find_weight(10, 5, 3)
    with pytest.raises(ValueError):
        oracle.find_weight(1.0, 1.0, 30.0)


def test_find_weight_value(oracle):
    # find_weight(10, 5, 3) = 13.2158...
    assert oracle.find_weight(10, 5, 3) == pytest.approx(13.215838362577491, rel=1e-15)


def _conv_cases():
    g = np.load(GC)
    return sorted({n.split('__')[0] for n in g.files if not n.startswith('complex')})


@pytest.mark.parametrize('name', _conv_cases())
def test_convolve_golden_bit_exact(oracle, name):
    g = np.load(GC)
    got = oracle.convolve(g[name + '__in'], g[name + '__k'], mode=str(g[name + '__mode']),
                          cval=float(g[name + '__cval']))
    np.testing.assert_array_equal(got, g[name + '__out'])


@pytest.mark.parametrize('mode', ['reflect', 'constant', 'nearest', 'mirror', 'wrap'])
def test_convolve_live_scipy(oracle, mode):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(17)
    for dt in (np.float32, np.float64):
        for shape, kshape in [((5, 14, 15), (1, 5, 3)), ((12, 9), (4, 6)), ((3, 3), (7, 7)), ((2, 5, 6, 7), (1, 3, 3, 3))]:
            a = rng.normal(size=shape).astype(dt)
            k = rng.normal(size=kshape)
            np.testing.assert_array_equal(oracle.convolve(a, k, mode=mode, cval=0.75),
                                          ndi.convolve(a, k, mode=mode, cval=0.75))


def test_extend_index_tables(oracle):
    """Border index maps against numpy.pad, which documents the same five conventions."""
    n = 5
    base = np.arange(n)
    for mode, npmode in [('reflect', 'symmetric'), ('mirror', 'reflect'), ('wrap', 'wrap'), ('nearest', 'edge')]:
        padded = np.pad(base, 13, mode=npmode)
        got = [oracle.lib().oracle_extend(i, n, {'reflect': 0, 'mirror': 3, 'wrap': 4, 'nearest': 2}[mode])
               for i in range(-13, n + 13)]
        np.testing.assert_array_equal(got, padded)
