"""GPU parity of the convolution HIP path (through the C ABI) against scipy.ndimage.convolve
(golden vectors + live scipy) and the CPU oracle.  The reference's own tests demand bit equality
with scipy (nd/tests/test_convolution_filter.py:39-47)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'convolve_scipy.npz')


def _gpu_convolve(a, k, device, **kw):
    import torch
    from nd_amd import kernels
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    out = kernels.convolve(t, k, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _cases():
    g = np.load(GOLD)
    return sorted({n.split('__')[0] for n in g.files if not n.startswith('complex')})


@pytest.mark.parametrize('name', _cases())
def test_golden_bit_exact(device, name):
    g = np.load(GOLD)
    a, k, want = g[name + '__in'], g[name + '__k'], g[name + '__out']
    got = _gpu_convolve(a, k, device, mode=str(g[name + '__mode']), cval=float(g[name + '__cval']))
    np.testing.assert_array_equal(got, want)


def test_golden_complex(device):
    """nd/filters.py:261-265: real and imaginary parts are convolved separately."""
    g = np.load(GOLD)
    z, k, want = g['complex64__in'], g['complex64__k'], g['complex64__out']
    re = _gpu_convolve(np.real(z), k, device)
    im = _gpu_convolve(np.imag(z), k, device)
    np.testing.assert_array_equal(re + 1j * im, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('mode', ['reflect', 'constant', 'nearest', 'mirror', 'wrap'])
def test_live_scipy_random(device, dtype, mode):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(3)
    for shape, kshape in [((6, 37, 41), (1, 5, 5)), ((3, 9, 130), (1, 3, 7)), ((2, 2, 17, 19), (1, 1, 4, 6)),
                          ((33, 65), (7, 3)), ((5, 4, 3), (3, 3, 3)), ((1, 1), (3, 3)), ((7,), (5,))]:
        a = rng.normal(size=shape).astype(dtype)
        k = rng.normal(size=kshape)
        want = ndi.convolve(a, k, mode=mode, cval=-1.25)
        got = _gpu_convolve(a, k, device, mode=mode, cval=-1.25)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('mode', ['reflect', 'constant', 'nearest', 'mirror', 'wrap'])
def test_three_dimensional_windows(device, dtype, mode):
    """Kernels with an extent along time as well (the reference hands scipy N-D kernels as they are,
    nd/filters.py:256-267): the tiled kernel walks the window's planes per output plane with the sums kept
    across them, scipy's plane-major footprint order -- bit-equal to scipy, every border mode, boxcars,
    even sizes, absent taps, a series shorter than the window's reach, a leading variable axis."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(11)
    cases = [((6, 37, 141), (3, 3, 3)), ((9, 40, 130), (5, 3, 5)), ((4, 33, 70), (3, 7, 7)), ((12, 20, 40), (5, 1, 1)),
             ((7, 35, 66), (2, 4, 4)), ((2, 30, 50), (3, 5, 3)), ((3, 3, 8, 150), (1, 3, 3, 3)), ((1, 17, 19), (3, 3, 3)),
             ((5, 64, 129), (7, 5, 5)),
             # windows along time or y alone, and pairs without x, on arrays large enough that the wrapper
             # used to transpose them to the back first: filtered where they lie
             ((6, 128, 256), (3, 1, 1)), ((6, 128, 256), (1, 5, 1)), ((6, 130, 250), (3, 5, 1)),
             ((5, 128, 256), (3, 1, 5)), ((2, 3, 128, 130), (1, 3, 1, 1))]
    for shape, kshape in cases:
        a = rng.normal(size=shape).astype(dtype)
        for k in (rng.normal(size=kshape), np.ones(kshape) / float(np.prod(kshape))):
            want = ndi.convolve(a, k, mode=mode, cval=0.75)
            got = _gpu_convolve(a, k, device, mode=mode, cval=0.75)
            np.testing.assert_array_equal(got, want, err_msg='%s %s' % (shape, kshape))
    a = rng.normal(size=(6, 21, 45)).astype(dtype)
    k = rng.normal(size=(3, 3, 5))
    k[1, 1, 2] = 0.0                   # scipy drops zero taps from the footprint
    k[2, 0, 4] = 0.0
    k1 = np.zeros((2, 1, 1)); k1[0, 0, 0] = 0.5           # one tap, one plane off the output's own
    k2 = np.zeros((3, 2, 1)); k2[0, 1, 0] = 2.0; k2[2, 0, 0] = -1.0      # the middle plane absent
    for kk in (k1, k1[::-1], k2):
        np.testing.assert_array_equal(_gpu_convolve(a, kk, device, mode=mode, cval=0.75),
                                      ndi.convolve(a, kk, mode=mode, cval=0.75))
    np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode, cval=0.75), ndi.convolve(a, k, mode=mode, cval=0.75))
    for origin in ((1, 0, -1), (-1, 1, 0)):
        np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode, cval=0.75, origin=origin),
                                      ndi.convolve(a, k, mode=mode, cval=0.75, origin=origin))


def test_origin_and_zero_taps(device):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(4)
    a = rng.normal(size=(20, 23)).astype(np.float32)
    k = rng.normal(size=(5, 4))
    k[1, 2] = 0.0                      # scipy drops |w| <= eps taps from the footprint
    k[3, 0] = 1e-17
    for origin in (0, (1, -1), (-2, 1)):
        want = ndi.convolve(a, k, origin=origin)
        got = _gpu_convolve(a, k, device, origin=origin)
        np.testing.assert_array_equal(got, want)


def test_strided_views_and_oracle(oracle, device):
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(8)
    base = rng.normal(size=(4, 30, 34)).astype(np.float32)
    k = np.ones((1, 5, 5)) / 25.0
    t = torch.from_numpy(base).to(device)
    view = t.permute(1, 2, 0)                          # (y, x, time) view of planar memory
    out = torch.empty_like(view)
    kernels.convolve(view, k.transpose(1, 2, 0), out=out)
    torch.cuda.synchronize()
    want = oracle.convolve(np.ascontiguousarray(base.transpose(1, 2, 0)), k.transpose(1, 2, 0))
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_large_footprint_uses_device_taps(device):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(9)
    a = rng.normal(size=(40, 44)).astype(np.float64)
    k = rng.normal(size=(13, 11))                      # 143 taps > 128
    np.testing.assert_array_equal(_gpu_convolve(a, k, device), ndi.convolve(a, k))


def test_full_size_properties(device):
    """Config-size checks that need no oracle: a normalised boxcar leaves a constant raster
    unchanged (to rounding), is linear, and the identity kernel is exact."""
    import torch
    from nd_amd import kernels
    t = torch.full((4, 2048, 2048), 3.25, device=device)
    k = np.ones((1, 5, 5)) / 25.0
    out = kernels.convolve(t, k)
    assert float((out - 3.25).abs().max()) < 1e-6
    x = torch.randn((2, 1024, 1024), device=device)
    ident = np.zeros((1, 3, 3)); ident[0, 1, 1] = 1
    assert torch.equal(kernels.convolve(x, ident), x)
    k3 = np.arange(9, dtype=float).reshape(1, 3, 3)
    a = kernels.convolve(x, k3)
    b = kernels.convolve(x * 2, k3)
    assert float((b - 2 * a).abs().max()) == 0.0      # exact: scaling by 2 commutes with rounding


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_reference_layout_goes_through_transpose_path(device, dtype):
    """(y, x, time) arrays with a (w, w, 1) kernel -- the reference's own layout -- are transposed
    on the device for the tiled kernel; the result must still be bit-equal to scipy."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(21)
    a = rng.normal(size=(300, 260, 3)).astype(dtype)           # > 2^16 elements
    for k in (np.ones((5, 5, 1)) / 25.0, rng.normal(size=(3, 7, 1)), rng.normal(size=(4, 3, 1))):
        np.testing.assert_array_equal(_gpu_convolve(a, k, device), ndi.convolve(a, k))
    b = rng.normal(size=(260, 2, 300)).astype(dtype)           # window over axes 0 and 2
    k = rng.normal(size=(5, 1, 3))
    np.testing.assert_array_equal(_gpu_convolve(b, k, device, mode='mirror'),
                                  ndi.convolve(b, k, mode='mirror'))


def test_tiled_edges_and_tiny_planes(device):
    """Tile borders (128 x 32 tiles), planes smaller than the window, all separable modes."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(22)
    for shape in [(2, 33, 129), (1, 65, 257), (3, 2, 3), (1, 1, 300), (2, 31, 127)]:
        a = rng.normal(size=shape).astype(np.float32)
        for kshape in [(1, 5, 5), (1, 11, 3), (1, 1, 7), (1, 9, 1), (1, 15, 15), (1, 3, 13)]:
            k = rng.normal(size=kshape)
            for mode in ('reflect', 'nearest', 'mirror', 'wrap'):
                np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode),
                                              ndi.convolve(a, k, mode=mode))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_windows_of_17_to_31_in_the_tiled_kernel(device, dtype):
    """Windows beyond 15 x 15 (round 6: BoxcarFilter(w=17) took the per-element kernel, 30 x the time of w = 15):
    every width class 17 .. 31, boxcars and random weights, taller-than-15 windows of few columns (run 17 wide with
    absent columns), even sizes, a zero tap, a 2 x 17 x 17 window along time, `constant` mode, tile borders and planes
    smaller than the window -- bit-equal to scipy."""
    import scipy.ndimage as ndi
    from nd_amd import kernels
    rng = np.random.default_rng(317)
    shapes = [(2, 70, 300), (1, 33, 129), (1, 9, 20)]
    for shape in shapes:
        a = rng.normal(size=shape).astype(dtype)
        for kshape in [(1, 17, 17), (1, 19, 19), (1, 21, 21), (1, 23, 23), (1, 25, 27), (1, 29, 29), (1, 31, 31),
                       (1, 21, 5), (1, 5, 21), (1, 17, 1), (1, 18, 18), (1, 16, 30)]:
            for k in (np.ones(kshape) / np.prod(kshape), rng.normal(size=kshape)):
                for mode in ('reflect', 'wrap'):
                    np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode), ndi.convolve(a, k, mode=mode),
                                                  err_msg=str((shape, kshape, mode)))
    a = rng.normal(size=(3, 40, 200)).astype(dtype)
    k = rng.normal(size=(1, 19, 21))
    k[0, 3, 4] = 0.0
    for mode, kw in (('constant', dict(cval=2.5)), ('nearest', {}), ('mirror', {})):
        np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode, **kw), ndi.convolve(a, k, mode=mode, **kw))
    k3 = rng.normal(size=(2, 17, 17))
    np.testing.assert_array_equal(_gpu_convolve(a, k3, device), ndi.convolve(a, k3))
    # the tiled kernel took them (not the per-element one): a 21 x 21 boxcar on 4 x 1024^2 in well under 5 ms
    import time
    import torch
    x = torch.from_numpy(rng.normal(size=(4, 1024, 1024)).astype(dtype)).to(device)
    kb = np.ones((1, 21, 21)) / 441.0
    kernels.convolve(x, kb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernels.convolve(x, kb)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5e-3


@pytest.mark.parametrize('mode', ['reflect', 'nearest', 'mirror', 'wrap'])
def test_register_window_kernel(device, mode):
    """Square 3 x 3 / 5 x 5 / 7 x 7 windows on float32 planes run in the register-window kernel
    (one wave per 256-column strip, DPP exchange of the edge columns): strip borders, widths that
    are not multiples of 4, planes shorter than the window, unaligned row pitches, random and
    boxcar weights, origins along y -- all bit-equal to scipy."""
    import scipy.ndimage as ndi
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(31)
    for shape in [(2, 70, 256), (1, 67, 257), (3, 5, 500), (2, 2, 9), (1, 130, 8), (2, 40, 1003), (1, 3, 767), (1, 9, 260), (2, 6, 252)]:
        a = rng.normal(size=shape).astype(np.float32)
        for w in (3, 5, 7):
            for k in (np.ones((1, w, w)) / float(w * w), rng.normal(size=(1, w, w))):
                np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode),
                                              ndi.convolve(a, k, mode=mode))
    a = rng.normal(size=(2, 45, 300)).astype(np.float32)
    k = rng.normal(size=(1, 5, 5))
    for origin in ((0, 1, 0), (0, -2, 0)):
        np.testing.assert_array_equal(_gpu_convolve(a, k, device, mode=mode, origin=origin),
                                      ndi.convolve(a, k, mode=mode, origin=origin))
    # a view with an odd row pitch (no 16-byte accesses) and an output with another pitch
    base = torch.from_numpy(rng.normal(size=(2, 50, 263)).astype(np.float32)).to(device)
    view = base[:, 3:47, 1:260]
    obuf = torch.empty((2, 44, 261), dtype=torch.float32, device=device)
    out = obuf[:, :, :259]
    k = np.ones((1, 3, 3)) / 9.0
    kernels.convolve(view, k, out=out, mode=mode)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), ndi.convolve(view.cpu().numpy(), k, mode=mode))


def test_output_may_alias_input(device):
    """scipy filters into a temporary when output and input share memory; so do the wrappers."""
    import scipy.ndimage as ndi
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(61)
    a = rng.normal(size=(3, 70, 300)).astype(np.float32)
    k = np.ones((1, 3, 3)) / 9.0
    t = torch.from_numpy(a.copy()).to(device)
    kernels.convolve(t, k, out=t)
    np.testing.assert_array_equal(t.cpu().numpy(), ndi.convolve(a, k))
    t = torch.from_numpy(a.copy()).to(device)
    kernels.gaussian_filter(t, (0, 1, 1), out=t)
    np.testing.assert_array_equal(t.cpu().numpy(), ndi.gaussian_filter(a, (0, 1, 1)))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_even_widths_and_constant_mode_in_the_tiled_kernel(device, dtype):
    """BoxcarFilter(w=4) / (w=6), even random kernels and mode='constant' (nd/filters.py:226 forwards
    any scipy keyword) on planes larger than one 128 x 32 tile, with infinities and NaNs in the data:
    an absent tap must not contribute 0 * inf.  Bit-equal to scipy."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(21)
    a = rng.gamma(4.0, 0.25, (3, 150, 300)).astype(dtype)
    a[0, 10, 20] = np.inf
    a[1, 149, 299] = np.nan
    a[2, 0, 0] = -np.inf
    cases = [(np.ones((1, 4, 4)) / 16.0, {}), (np.ones((1, 6, 6)) / 36.0, {}),
             (rng.normal(size=(1, 4, 6)), {}), (rng.normal(size=(1, 5, 2)), {'mode': 'nearest'}),
             (np.ones((1, 5, 5)) / 25.0, {'mode': 'constant', 'cval': 0.0}),
             (rng.normal(size=(1, 3, 3)), {'mode': 'constant', 'cval': 2.5}),
             (rng.normal(size=(1, 4, 4)), {'mode': 'constant', 'cval': -1.0}),
             (rng.normal(size=(1, 7, 14)), {'mode': 'wrap'})]
    with np.errstate(all='ignore'):
        for k, kw in cases:
            want = ndi.convolve(a, k, **kw)
            got = _gpu_convolve(a, k, device, **kw)
            np.testing.assert_array_equal(got, want)
    # the reference's own layout: (y, x, time), window over the two leading axes
    b = np.ascontiguousarray(np.moveaxis(a, 0, -1))
    with np.errstate(all='ignore'):
        for k, kw in cases[:2] + cases[4:6]:
            k3 = np.moveaxis(k, 0, -1)
            np.testing.assert_array_equal(_gpu_convolve(b, k3, device, **kw), ndi.convolve(b, k3, **kw))
