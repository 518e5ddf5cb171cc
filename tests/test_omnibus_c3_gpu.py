"""Full-pol (C3) omnibus on the GPU against this repository's generic-p oracle.  The reference has
no p = 3 implementation, so this is self-consistency (HIP vs C restatement vs the float64 numpy
definition in tests/test_oracle_omnibus.py), not reference parity."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _run(planes, alpha, n, device, layout='tyx'):
    import torch
    from nd_amd import kernels
    ts = [torch.from_numpy(p).to(device) for p in planes]
    dims = ('time', 'y', 'x')
    if layout == 'yxt':
        ts = [t.permute(1, 2, 0).contiguous() for t in ts]
        dims = ('y', 'x', 'time')
    ch, z, P = kernels.change_detection_c3(ts, alpha=alpha, n=n, dims=dims, stats=True)
    ch2 = kernels.change_detection_c3(ts, alpha=alpha, n=n, dims=dims)
    torch.cuda.synchronize()
    assert torch.equal(ch, ch2)
    return ch.cpu().numpy(), z.cpu().numpy(), P.cpu().numpy()


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('k', [7, 12, 48])
@pytest.mark.parametrize('alpha', [0.9, 0.99])
def test_c3_against_generic_oracle(oracle, device, dtype, k, alpha):
    planes = synth.omnibus_stack_c3(seed=k, k=k, ny=20, nx=70, dtype=dtype, change_frac=0.15)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    want, z0, P0 = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8, stats=True)
    ch, z, P = _run(planes, alpha, 9, device)
    assert int((ch != want).sum()) == 0
    np.testing.assert_allclose(z, z0, rtol=1e-5, equal_nan=True)
    np.testing.assert_allclose(P, P0, rtol=1e-5, atol=1e-30, equal_nan=True)
    assert want.sum() > 0


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('k', [8, 12, 16, 32, 48, 64, 96])
def test_c3_pixel_major_entry_point(oracle, device, dtype, k):
    """nd_amd_omnibus_c3_pixel_major: the nine variables in the reference's (y, x, time) layout read where
    they lie -- nine real arrays, or three real and three interleaved complex ones; ragged rasters; z / P
    rasters; the sparse regime.  Against the generic-p oracle and the planar entry point."""
    import torch
    from nd_amd import kernels
    if dtype == np.float64 and k >= 64:
        pytest.skip('9 k doubles of 16 pixels exceed the images')
    for ny, nx in [(1, 5), (7, 70), (12, 131)]:
        planes = synth.omnibus_stack_c3(seed=k + nx, k=k, ny=ny, nx=nx, dtype=dtype, change_frac=0.2)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        dev = [torch.from_numpy(a).to(device) for a in yxt]
        cplx = [torch.complex(dev[c], dev[c + 1]) for c in (3, 5, 7)]
        joint = dev[:3] + [h for z_ in cplx for h in (z_.real, z_.imag)]
        for alpha in (0.8, 0.99):
            want, z0, P0 = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8, stats=True)
            got = kernels.change_detection_c3_pixel_major(dev, alpha=alpha, n=9)
            assert got is not None
            np.testing.assert_array_equal(got.cpu().numpy(), want)
            res = kernels.change_detection_c3_pixel_major(joint, alpha=alpha, n=9, stats=True)
            assert res is not None
            np.testing.assert_array_equal(res[0].cpu().numpy(), want)
            np.testing.assert_allclose(res[1].cpu().numpy(), z0, rtol=1e-5, equal_nan=True)
            np.testing.assert_allclose(res[2].cpu().numpy(), P0, rtol=1e-5, atol=1e-30, equal_nan=True)
            planar = kernels.change_detection_c3([torch.from_numpy(p).to(device) for p in planes], alpha=alpha, n=9)
            assert torch.equal(planar, got)
        # below the sparse regime: declined (the caller transposes)
        assert kernels.change_detection_c3_pixel_major(dev, alpha=0.01, n=9) is None
    t = torch.ones((3, 4, 10), device=device)                      # 10 dates: not whole 16-byte vectors
    assert kernels.change_detection_c3_pixel_major([t] * 9, alpha=0.9) is None


def test_c3_layouts_and_tails(oracle, device):
    planes = synth.omnibus_stack_c3(seed=3, k=6, ny=5, nx=131, dtype=np.float32, change_frac=0.3)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    want = oracle.change_detection_pol(yxt, 3, 0.9, 4, njobs=4)
    for layout in ('tyx', 'yxt'):
        ch, _, _ = _run(planes, 0.9, 4, device, layout)
        np.testing.assert_array_equal(ch, want)


def test_omnibus_test_class_full_pol_is_opt_in(oracle, device):
    """OmnibusTest(pol='full').apply on a dataset with C33 / C13 / C23 runs the 3 x 3 test (host and
    device datasets, complex cross terms); without `pol` the same dataset gets the reference's
    dual-pol test on C11 / C12 / C22 (nd/change.py:66), C33 ignored."""
    import torch
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    rng = np.random.default_rng(81)
    k, ny, nx, looks = 8, 12, 70, 9
    s = (rng.normal(size=(3, looks, k, ny, nx)) + 1j * rng.normal(size=(3, looks, k, ny, nx))) / np.sqrt(2)
    gain = np.where((np.arange(k)[:, None, None] >= 4) & (rng.random((ny, nx)) < 0.4)[None], 3.0, 1.0)
    s = s * np.sqrt(gain)[None, None]
    cov = lambda i, j: (s[i] * np.conj(s[j])).mean(axis=0)                     # (k, ny, nx)
    yxt = lambda a: np.ascontiguousarray(np.moveaxis(a, 0, -1))
    host = xr_lite.Dataset()
    for i, name in enumerate(('C11', 'C22', 'C33')):
        host[name] = (('y', 'x', 'time'), yxt(cov(i, i).real.astype(np.float32)))
    for (i, j), name in (((0, 1), 'C12'), ((0, 2), 'C13'), ((1, 2), 'C23')):
        host[name] = (('y', 'x', 'time'), yxt(cov(i, j).astype(np.complex64)))
    planes = [host['C11'].values, host['C22'].values, host['C33'].values]
    for name in ('C12', 'C13', 'C23'):
        planes += [np.ascontiguousarray(host[name].values.real), np.ascontiguousarray(host[name].values.imag)]
    want = oracle.change_detection_pol(planes, 3, 0.9, looks, njobs=4).astype(bool)
    assert want.any()
    got = OmnibusTest(n=looks, alpha=0.9, pol='full').apply(host)
    assert isinstance(got.values, np.ndarray) and got.dims == ('y', 'x', 'time')
    np.testing.assert_array_equal(got.values, want)
    dev_ds = xr_lite.Dataset()
    for name in host.data_vars:
        dev_ds[name] = (('y', 'x', 'time'), torch.from_numpy(host[name].values).to(device))
    got_dev = OmnibusTest(n=looks, alpha=0.9, pol='full').apply(dev_ds)
    np.testing.assert_array_equal(got_dev.values.cpu().numpy(), want)
    # default: dual-pol on C11 / C12 / C22, as upstream; also when C13 / C23 are absent
    c12 = host['C12'].values
    dual = [host['C11'].values, np.ascontiguousarray(c12.real), np.ascontiguousarray(c12.imag),
            host['C22'].values]
    want2 = oracle.change_detection_planes(dual, 0.9, looks, njobs=4).astype(bool)
    np.testing.assert_array_equal(OmnibusTest(n=looks, alpha=0.9).apply(host).values, want2)
    del host['C13'], host['C23']
    np.testing.assert_array_equal(OmnibusTest(n=looks, alpha=0.9).apply(host).values, want2)
    with pytest.raises(KeyError):
        OmnibusTest(n=looks, alpha=0.9, pol='full').apply(host)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('k', [2, 3, 5, 24, 48, 64, 65, 96])
def test_c3_low_thresholds_streaming_search(oracle, device, dtype, k):
    """The thresholds users pass (the reference's default 0.01, the tutorial's 1e-4, and a middle
    one) run the fused streaming search (omnibus_c3_stream_kernel, up to 96 dates): same map as the
    generic-p oracle byte for byte, planar and strided, with and without the z / P rasters."""
    planes = synth.omnibus_stack_c3(seed=100 + k, k=k, ny=12 if k <= 64 else 4, nx=140, dtype=dtype, change_frac=0.3)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    for alpha in (1e-4, 0.01, 0.3):
        want, z0, P0 = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8, stats=True)
        for layout in ('tyx', 'yxt'):
            ch, z, P = _run(planes, alpha, 9, device, layout)
            assert int((ch != want).sum()) == 0, (k, alpha, layout)
            np.testing.assert_allclose(z, z0, rtol=1e-5, equal_nan=True)
            np.testing.assert_allclose(P, P0, rtol=1e-5, atol=1e-30, equal_nan=True)
        if k > 2:
            assert want.sum() > 0


def test_c3_streaming_search_degenerate_values(oracle, device):
    """Zeros, negatives, NaN, infinities, non-positive-definite dates and tiny / huge magnitudes:
    whatever the float32 screen cannot vouch for goes to the exact pass, so the map still equals
    the oracle's."""
    rng = np.random.default_rng(9)
    planes = synth.omnibus_stack_c3(seed=77, k=16, ny=10, nx=130, dtype=np.float32, change_frac=0.3)
    planes = [p.copy() for p in planes]
    for val in (0.0, -1.0, np.nan, np.inf):
        m = rng.random(planes[0].shape) < 0.004
        planes[int(rng.integers(0, 9))][m] = val
    planes[3][:, 2, 10:40] *= 5.0                      # |C12|^2 > C11 C22: not positive semi-definite
    for p in planes:
        p[:, 5, :] *= 1e-12                            # determinants ~1e-36
        p[:, 6, :] *= 3e9
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    for alpha in (0.01, 0.5):
        with np.errstate(all='ignore'):
            want = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8)
        ch, _, _ = _run(planes, alpha, 9, device)
        assert int((ch != want).sum()) == 0, alpha


def test_c3_nodata_margins(oracle, device):
    """NaN / zero fill and single bad dates under the full-pol streaming search: no change, no
    exact pass, same map as the oracle."""
    planes = [p.copy() for p in synth.omnibus_stack_c3(seed=88, k=20, ny=8, nx=200, dtype=np.float32, change_frac=0.3)]
    for p in planes:
        p[:, :, 0:40] = np.nan
        p[:, :, 40:80] = 0.0
        p[4, :, 80:100] = 0.0
    planes[2][7, :, 100:120] = np.nan
    planes[0][0, :, 120:140] = np.inf
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    for alpha in (1e-4, 0.01, 0.5):
        with np.errstate(all='ignore'):
            want = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8)
        ch, _, _ = _run(planes, alpha, 9, device)
        assert int((ch != want).sum()) == 0, alpha
        assert not ch[:, 0:140].any() and ch[:, 140:].any()


@pytest.mark.parametrize('lanes', ['64', '32', '16'])
def test_c3_pass_b_pixels_per_wave(lanes):
    """ND_AMD_C3_LANES: 64 / 32 / 16 listed pixels per wave of the full-pol pass B (image size against
    waves per CU): same map as the generic-p oracle, in a fresh process per width."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from nd_amd import kernels, synth\n"
        "from oracle import oracle as O\n"
        "O.build()\n"
        "for k, alpha in ((48, 0.99), (20, 0.9), (9, 0.99)):\n"
        "    st = synth.wishart_c3_stack(k, 40, 333, looks=9, seed=k, device='cuda', change_frac=0.2)\n"
        "    got = kernels.change_detection_c3([st[c] for c in range(9)], alpha=alpha, n=9).cpu().numpy()\n"
        "    host = st.cpu().numpy()\n"
        "    want = O.change_detection_pol([np.moveaxis(host[c], 0, -1) for c in range(9)], 3, alpha, 9, njobs=8)\n"
        "    assert np.array_equal(got, want), (k, alpha, int((got != want).sum()))\n"
        "    assert want.sum() > 0\n"
        "print('ok')\n" % root)
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, ND_AMD_C3_LANES=lanes),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize('k', [5, 16, 33, 48, 64])
def test_c3_sparse_regime_degenerate_values(oracle, device, k):
    """The time-split pass A of the sparse regime (omnibus_c3_retain_kernel: four waves share a pixel's
    time axis, candidates dumped from registers, the dump searched one lane per pixel) on values its
    re-associated screen cannot vouch for: non-positive-definite dates, under- and overflowing products
    of determinants, prefix products far from the suffix's, nodata -- all must reach the exact pass or
    be provably silent; more candidates than the dump holds; ragged rows."""
    rng = np.random.default_rng(k)
    planes = [p.copy() for p in synth.omnibus_stack_c3(seed=900 + k, k=k, ny=6, nx=333, dtype=np.float32,
                                                       change_frac=0.3)]
    planes[3][:, 0, 10:40] *= 5.0                      # |C12|^2 > C11 C22
    planes[0][:, 1, 5:25] *= -1.0
    for p in planes:
        p[:, 2, 0:60] *= 1e-8                          # determinants ~1e-24: the product underflows for long series
        p[:, 2, 60:120] *= 3e6
        p[k // 2:, 3, 0:50] *= 1e-6
        p[:, 3, 100:130] = 0.0
        p[:, 3, 130:160] = np.nan
    planes[2][k - 1, 4, 0:30] = np.inf
    planes[5][0, 4, 30:60] = np.nan
    for val in (0.0, -1.0):
        m = rng.random(planes[0].shape) < 0.002
        planes[int(rng.integers(0, 9))][m] = val
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    import torch
    from nd_amd import kernels
    dev = [torch.from_numpy(p).to(device) for p in planes]
    for alpha in (0.8, 0.99):
        with np.errstate(all='ignore'):
            want = oracle.change_detection_pol(yxt, 3, alpha, 9, njobs=8)
        got = kernels.change_detection_c3(dev, alpha=alpha, n=9)
        torch.cuda.synchronize()
        assert int((got.cpu().numpy() != want).sum()) == 0, (k, alpha)
        assert want.sum() > 0
