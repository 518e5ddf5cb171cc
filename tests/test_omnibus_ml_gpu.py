"""OmnibusTest(ml=w) with the multilooking fused into the test (nd_amd_omnibus_c2_ml) against
(i) the CPU oracle's  scipy boxcar -> change_detection  (nd/change.py:61-69: BoxcarFilter(w=ml) then
n = ml ** 2) and (ii) the library's own two-step path (nd_amd_correlate on every plane, then
nd_amd_omnibus_c2) -- bit for bit: change map, z and P rasters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stack(seed, k, ny, nx, looks=1, change_frac=0.05, scale=1.0):
    from tests import synth
    planes = synth.omnibus_stack(seed, k, ny, nx, looks=looks, dtype=np.float32, change_frac=change_frac)
    return [np.ascontiguousarray(p * np.float32(scale)) for p in planes]          # (time, y, x) each


def _oracle_ml(oracle, planes, ml, alpha, stats=False):
    import scipy.ndimage as ndi
    kern = (np.ones((ml, ml), dtype=np.float64) / ml ** 2).reshape(1, ml, ml)
    mlp = [ndi.convolve(p, kern) for p in planes]                                   # nd/filters.py:262-267
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in mlp]
    with np.errstate(all='ignore'):
        return oracle.change_detection_planes(yxt, alpha, ml * ml, njobs=8, stats=stats), mlp


def _two_step(kernels, torch, dev_planes, ml, alpha, stats=False):
    stack = torch.stack(dev_planes)
    kern = (np.ones((ml, ml), dtype=np.float64) / ml ** 2).reshape(1, 1, ml, ml)
    mlk = kernels.convolve(stack, kern)
    return kernels.change_detection(mlk[0], mlk[1], mlk[2], mlk[3], alpha=alpha, n=ml * ml, stats=stats)


SHAPES = [(24, 13, 70), (24, 37, 129), (24, 12, 64), (24, 25, 300), (8, 40, 200), (9, 17, 65),
          (16, 31, 127), (17, 50, 90), (2, 30, 140), (3, 14, 64), (7, 6, 6), (24, 4, 500), (23, 64, 193)]


@pytest.mark.parametrize('ml', [3, 5])
@pytest.mark.parametrize('alpha', [1e-4, 0.01, 0.3, 0.9, 0.99])
def test_fused_multilook_equals_oracle_and_two_step(oracle, device, ml, alpha):
    import torch
    from nd_amd import kernels
    for i, (k, ny, nx) in enumerate(SHAPES):
        if ny <= ml - 1 or nx <= ml - 1:
            continue
        planes = _stack(100 + i, k, ny, nx)
        dev = [torch.from_numpy(p).to(device) for p in planes]
        got = kernels.change_detection_multilooked(*dev, alpha=alpha, ml=ml)
        assert got is not None, 'fused kernel refused %s' % ((k, ny, nx),)
        want, _ = _oracle_ml(oracle, planes, ml, alpha)
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg='shape %s' % ((k, ny, nx),))
        two = _two_step(kernels, torch, dev, ml, alpha)
        assert torch.equal(got, two)


@pytest.mark.parametrize('ml', [3, 5])
def test_fused_multilook_statistics(oracle, device, ml):
    import torch
    from nd_amd import kernels
    planes = _stack(7, 24, 45, 210)
    dev = [torch.from_numpy(p).to(device) for p in planes]
    for alpha in (0.01, 0.99):
        ch, z, P = kernels.change_detection_multilooked(*dev, alpha=alpha, ml=ml, stats=True)
        (cw, zw, pw), _ = _oracle_ml(oracle, planes, ml, alpha, stats=True)
        np.testing.assert_array_equal(ch.cpu().numpy(), cw)
        np.testing.assert_allclose(z.cpu().numpy(), zw, rtol=1e-5, atol=0, equal_nan=True)
        np.testing.assert_allclose(P.cpu().numpy(), pw, rtol=1e-5, atol=1e-7, equal_nan=True)
        c2, z2, P2 = _two_step(kernels, torch, dev, ml, alpha, stats=True)
        assert torch.equal(ch, c2)
        assert torch.equal(z.view(torch.int32), z2.view(torch.int32))
        assert torch.equal(P.view(torch.int32), P2.view(torch.int32))


def test_fused_multilook_degenerate_values(oracle, device):
    """zeros (nodata), NaN, inf and negative determinants inside the windows."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(5)
    planes = _stack(11, 24, 40, 150)
    planes[0][:, :10, :] = 0
    planes[3][:, :10, :] = 0
    planes[1][:, :10, :] = 0
    planes[2][:, :10, :] = 0
    planes[0][3, 20, 30] = np.nan
    planes[3][7, 25, 100] = np.inf
    planes[1][5, 30:33, 60:70] *= 50                       # |C12|^2 > C11 C22: negative determinants
    for v in planes:
        v[rng.integers(0, 24), rng.integers(0, 40), rng.integers(0, 150)] = 0
    dev = [torch.from_numpy(p).to(device) for p in planes]
    for ml in (3, 5):
        for alpha in (0.01, 0.5, 0.99):
            got = kernels.change_detection_multilooked(*dev, alpha=alpha, ml=ml)
            want, _ = _oracle_ml(oracle, planes, ml, alpha)
            np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_fused_multilook_padded_planes_and_scales(oracle, device):
    """date planes with a padded pitch (nd_amd.synth.empty_stack) and magnitudes away from 1."""
    import torch
    from nd_amd import kernels, synth
    for scale in (1e-4, 3e3):
        planes = _stack(21, 24, 33, 257, scale=scale)
        st = synth.empty_stack(4, 24, 33, 257, device, torch.float32)
        for i, p in enumerate(planes):
            st[i].copy_(torch.from_numpy(p))
        for alpha in (0.01, 0.99):
            got = kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=alpha, ml=3)
            assert got is not None
            want, _ = _oracle_ml(oracle, planes, 3, alpha)
            np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_apply_uses_the_fused_path(oracle, device, monkeypatch):
    """OmnibusTest(ml=3).apply on device data gives the oracle's map through the fused kernel, and
    the same map with ND_AMD_ML_FUSED=0 (two-step path)."""
    import torch
    from nd_amd import kernels, xr_lite
    from nd_amd.change import OmnibusTest
    planes = _stack(31, 24, 60, 200)
    calls = []
    orig = kernels.change_detection_multilooked

    def spy(*a, **kw):
        r = orig(*a, **kw)
        calls.append(r is not None)
        return r

    monkeypatch.setattr(kernels, 'change_detection_multilooked', spy)
    ds = xr_lite.Dataset()
    for v, p in zip(('C11', 'C12__re', 'C12__im', 'C22'), planes):
        ds[v] = (('time', 'y', 'x'), torch.from_numpy(p).to(device))
    want, _ = _oracle_ml(oracle, planes, 3, 0.9)
    got = OmnibusTest(ml=3, alpha=0.9).apply(ds)
    assert calls == []                       # below the sparse regime: boxcar kernel + fused search
    np.testing.assert_array_equal(got.values.cpu().numpy(), want.astype(bool))
    monkeypatch.setenv('ND_AMD_ML_FUSED', '2')
    got = OmnibusTest(ml=3, alpha=0.9).apply(ds)
    assert calls == [True]
    np.testing.assert_array_equal(got.values.cpu().numpy(), want.astype(bool))
    # reference layout (y, x, time) on the host
    dh = xr_lite.Dataset()
    for v, p in zip(('C11', 'C12__re', 'C12__im', 'C22'), planes):
        dh[v] = (('y', 'x', 'time'), np.ascontiguousarray(np.moveaxis(p, 0, -1)))
    got = OmnibusTest(ml=3, alpha=0.9).apply(dh)
    assert calls == [True, True]
    np.testing.assert_array_equal(got.values, want.astype(bool))
    monkeypatch.setenv('ND_AMD_ML_FUSED', '0')
    got = OmnibusTest(ml=3, alpha=0.9).apply(dh)
    assert calls == [True, True]
    np.testing.assert_array_equal(got.values, want.astype(bool))
    # the sparse regime takes the fused kernel by default
    monkeypatch.delenv('ND_AMD_ML_FUSED')
    want99, _ = _oracle_ml(oracle, planes, 3, 0.99)
    got = OmnibusTest(ml=3, alpha=0.99).apply(ds)
    assert calls == [True, True, True]
    np.testing.assert_array_equal(got.values.cpu().numpy(), want99.astype(bool))
