"""GPU parity of the omnibus HIP path (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _run_gpu(planes_np, alpha, n, device, layout='tyx'):
    import torch
    from nd_amd import kernels
    ts = [torch.from_numpy(p).to(device) for p in planes_np]
    if layout == 'tyx':
        dims = ('time', 'y', 'x')
    elif layout == 'yxt':
        ts = [t.permute(1, 2, 0).contiguous() for t in ts]
        dims = ('y', 'x', 'time')
    elif layout == 'yxtv':
        # the reference's own in-memory form: one (y, x, time, 4) array
        stacked = torch.stack(ts, dim=-1).permute(1, 2, 0, 3).contiguous()
        ts = [stacked[..., v] for v in range(4)]
        dims = ('y', 'x', 'time')
    ch, z, P = kernels.change_detection(*ts, alpha=alpha, n=n, dims=dims, stats=True)
    torch.cuda.synchronize()
    return ch.cpu().numpy(), z.cpu().numpy(), P.cpu().numpy()


def _run_oracle(oracle, planes_np, alpha, n):
    planes = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes_np]   # (y, x, t)
    return oracle.change_detection_planes(planes, alpha, n, njobs=8, stats=True)


def _compare(got, want, rtol=1e-5):
    ch, z, P = got
    ch0, z0, P0 = want
    assert ch.shape == ch0.shape and ch.dtype == np.uint8
    nbad = int((ch != ch0).sum())
    assert nbad == 0, '%d change-map bytes differ' % nbad
    np.testing.assert_allclose(z, z0, rtol=rtol, atol=0, equal_nan=True)
    np.testing.assert_allclose(P, P0, rtol=rtol, atol=1e-30, equal_nan=True)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('alpha', [0.5, 0.9, 0.99, 0.9999])
def test_wishart_parity(oracle, device, dtype, alpha):
    planes = synth.omnibus_stack(seed=7, k=24, ny=96, nx=128, dtype=dtype)
    want = _run_oracle(oracle, planes, alpha, 9)
    got = _run_gpu(planes, alpha, 9, device)
    _compare(got, want)
    assert want[0].sum() > 0


@pytest.mark.parametrize('shape', [(24, 33, 77), (5, 8, 1030), (2, 3, 5), (37, 16, 64),
                                   (1, 4, 4), (24, 1, 1), (70, 9, 12)])
@pytest.mark.parametrize('layout', ['tyx', 'yxt', 'yxtv'])
def test_shapes_and_layouts(oracle, device, shape, layout):
    k, ny, nx = shape
    planes = synth.omnibus_stack(seed=k * 1000 + nx, k=k, ny=ny, nx=nx, dtype=np.float32,
                                 change_frac=0.2)
    want = _run_oracle(oracle, planes, 0.9, 9)
    got = _run_gpu(planes, 0.9, 9, device, layout)
    _compare(got, want)


@pytest.mark.parametrize('n', [1, 4, 9, 25])
def test_number_of_looks(oracle, device, n):
    planes = synth.omnibus_stack(seed=11, k=12, ny=40, nx=64, looks=max(n, 2), dtype=np.float32,
                                 change_frac=0.1)
    want = _run_oracle(oracle, planes, 0.95, n)
    got = _run_gpu(planes, 0.95, n, device)
    _compare(got, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_default_number_of_looks_on_multilooked_data(oracle, device, dtype):
    """n = 1 -- the reference's DEFAULT -- on nine-look data: omega2 of the whole-series test leaves [0, 1], no screen can
    decide it, and pass A evaluates it exactly instead (round 6; before, every pixel went through pass B: 14 x the time).
    Maps (with and without the rasters) equal the oracle's at every threshold and series length (the reference reports
    no change at all there: its P is negative), as do those of n = 2, 3, whose screens are usable or not by length."""
    import torch
    from nd_amd import kernels
    for k, ny, nx in ((3, 20, 70), (5, 33, 129), (24, 40, 300), (40, 12, 200), (70, 6, 130)):
        planes = synth.omnibus_stack(seed=300 + k, k=k, ny=ny, nx=nx, looks=9, dtype=dtype, change_frac=0.3)
        for n in (1, 2, 3):
            for alpha in (0.01, 0.5, 0.99):
                want = _run_oracle(oracle, planes, alpha, n)
                got = _run_gpu(planes, alpha, n, device)
                _compare(got, want)
                ts = [torch.from_numpy(np.ascontiguousarray(p)).to(device) for p in planes]
                ch = kernels.change_detection(*ts, alpha=alpha, n=n)
                torch.cuda.synchronize()
                assert np.array_equal(ch.cpu().numpy(), want[0]), (k, n, alpha)
    # (with n = 1 the reference's P is negative everywhere on such data -- no change at any length; n = 3 finds changes)
    planes = synth.omnibus_stack(seed=303, k=3, ny=20, nx=70, looks=9, dtype=dtype, change_frac=0.3)
    assert _run_oracle(oracle, planes, 0.01, 1)[0].sum() == 0 and _run_oracle(oracle, planes, 0.01, 3)[0].sum() > 0


def test_reference_known_answer(oracle, device):
    """nd/tests/test_change_omnibus.py:6-19."""
    dims = {'y': 5, 'x': 5, 'time': 10}
    d1 = synth.reference_test_dataset(dims, [1, 0, 0, 1], 0.1)
    d2 = synth.reference_test_dataset(dims, [10, 0, 0, 10], 0.1)
    ds = {v: np.concatenate([d1[v][..., :5], d2[v][..., 5:]], axis=2) for v in d1}
    for dtype in (np.float64, np.float32):
        planes = [np.ascontiguousarray(np.moveaxis(ds[v], -1, 0)).astype(dtype)
                  for v in ('C11', 'C12__re', 'C12__im', 'C22')]
        ch, z, P = _run_gpu(planes, 0.9, 9, device)
        assert ch[:, :, 5].all()
        assert (ch.sum(axis=2) == 1).all()
        _compare((ch, z, P), _run_oracle(oracle, planes, 0.9, 9))


def test_nan_path(oracle, device):
    """nd/tests/test_change_common.py:21-32: N(0,1) data -> negative determinants -> NaN."""
    ds = synth.reference_test_dataset({'y': 20, 'x': 30, 'time': 10}, 0, 1)
    planes = [np.ascontiguousarray(np.moveaxis(ds[v], -1, 0))
              for v in ('C11', 'C12__re', 'C12__im', 'C22')]
    got = _run_gpu(planes, 0.01, 1, device)
    want = _run_oracle(oracle, planes, 0.01, 1)
    _compare(got, want)


def test_degenerate_values(oracle, device):
    """zeros (nodata), identical matrices, one zero date, infs."""
    k, ny, nx = 8, 4, 16
    planes = [p.copy() for p in synth.omnibus_stack(3, k, ny, nx, dtype=np.float32)]
    for p in planes:
        p[:, 0, 0] = 0                      # all-zero pixel
    planes[0][:, 0, 1] = 1; planes[1][:, 0, 1] = 0; planes[2][:, 0, 1] = 0; planes[3][:, 0, 1] = 1
    for p in planes:
        p[3, 0, 2] = 0                      # one zero date
    planes[0][2, 0, 3] = np.inf
    planes[0][2, 0, 4] = np.nan
    planes[0][:, 0, 5] *= 1e-30             # underflowing product
    planes[3][:, 0, 5] *= 1e-30
    planes[0][:, 0, 6] *= 1e18              # overflowing product
    planes[3][:, 0, 6] *= 1e18
    got = _run_gpu(planes, 0.5, 9, device)
    want = _run_oracle(oracle, planes, 0.5, 9)
    _compare(got, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_minimal_workspace_gathers_from_planes(oracle, device, dtype):
    """Without room for the series dump every listed pixel is gathered from the planes by the
    search kernel; also n = 1 / small alpha, where the decision bounds are disabled (omega2 > 1)
    and every pixel is listed."""
    import torch
    from nd_amd import kernels
    planes = synth.omnibus_stack(seed=17, k=9, ny=30, nx=90, looks=2, dtype=dtype, change_frac=0.3)
    ts = [torch.from_numpy(p).to(device) for p in planes]
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    for alpha, n in [(0.9, 9), (0.01, 1), (0.5, 1)]:
        want = oracle.change_detection_planes(yxt, alpha, n, njobs=8)
        for ws in ('minimal', 'recommended'):
            got = kernels.change_detection(*ts, alpha=alpha, n=n, workspace=ws)
            np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('dtype,k', [(np.float32, 24), (np.float32, 7), (np.float32, 16), (np.float64, 12),
                                     (np.float64, 5), (np.float32, 21), (np.float32, 22), (np.float32, 23),
                                     (np.float32, 10), (np.float32, 2), (np.float32, 3), (np.float64, 11)])
def test_pixel_major_kernel_equals_planar(oracle, device, dtype, k):
    """nd_amd_omnibus_c2_pixel_major: variables in the reference's (y, x, time) layout, C12 as one
    interleaved complex tensor or as two real ones; ragged raster sizes (last spans of fewer than 64 pixels);
    z / P output; every threshold regime; series lengths that are and are not a multiple of the 16-byte vector
    (round 6: the latter read the LDS image element by element instead of taking the register-staged kernel);
    equal to the oracle and to the planar entry point."""
    import torch
    from nd_amd import kernels
    from tests import synth as tsynth
    for ny, nx in [(1, 1), (3, 70), (33, 257), (64, 64)]:
        planes = tsynth.omnibus_stack(seed=ny * nx + k, k=k, ny=ny, nx=nx, dtype=dtype, change_frac=0.3)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        dev = [torch.from_numpy(a).to(device) for a in yxt]
        c12 = torch.complex(dev[1], dev[2])
        for alpha in (0.9, 0.01, 0.3):
            want, zw, pw = oracle.change_detection_planes(yxt, alpha, 9, njobs=4, stats=True)
            got = kernels.change_detection_pixel_major(dev[0], dev[1], dev[2], dev[3], alpha=alpha, n=9)
            assert got is not None
            np.testing.assert_array_equal(got.cpu().numpy(), want)
            got = kernels.change_detection_pixel_major(dev[0], c12.real, c12.imag, dev[3], alpha=alpha, n=9)
            np.testing.assert_array_equal(got.cpu().numpy(), want)
            res = kernels.change_detection_pixel_major(dev[0], c12.real, c12.imag, dev[3], alpha=alpha, n=9,
                                                       stats=True)
            np.testing.assert_array_equal(res[0].cpu().numpy(), want)
            np.testing.assert_allclose(res[1].cpu().numpy(), zw, rtol=1e-5, equal_nan=True)
            np.testing.assert_allclose(res[2].cpu().numpy(), pw, rtol=1e-5, atol=1e-7, equal_nan=True)
    # not that layout / too long a series: declined
    t = torch.zeros((4, 5, 30), device=device)
    assert kernels.change_detection_pixel_major(t, t, t, t, alpha=0.9) is None
    t = torch.zeros((6, 4, 5), device=device).permute(1, 2, 0)
    assert kernels.change_detection_pixel_major(t, t, t, t, alpha=0.9) is None


@pytest.mark.parametrize('dtype,k', [(np.float32, 28), (np.float32, 48), (np.float32, 96), (np.float32, 128),
                                     (np.float32, 192), (np.float64, 16), (np.float64, 48), (np.float64, 96)])
def test_pixel_major_long_series(oracle, device, dtype, k):
    """The reference's layout beyond the register-retaining sizes (nd/change.py:66-67 hands the native code
    (y, x, time) arrays of any length): in the sparse regime the pixel-major entry point folds the series
    out of LDS images (64, 32 or 16 pixels per wave) and the search reads the listed series in place.
    Interleaved complex C12 and two real arrays, ragged rasters, z / P rasters; below the sparse regime and
    for lengths that are not whole 16-byte vectors the entry point declines (the caller transposes)."""
    import torch
    from nd_amd import kernels
    from tests import synth as tsynth
    for ny, nx in [(1, 5), (7, 70), (20, 131)]:
        planes = tsynth.omnibus_stack(seed=ny + nx + k, k=k, ny=ny, nx=nx, dtype=dtype, change_frac=0.3)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        dev = [torch.from_numpy(a).to(device) for a in yxt]
        c12 = torch.complex(dev[1], dev[2])
        for alpha in (0.8, 0.99):
            want, zw, pw = oracle.change_detection_planes(yxt, alpha, 9, njobs=4, stats=True)
            got = kernels.change_detection_pixel_major(dev[0], dev[1], dev[2], dev[3], alpha=alpha, n=9)
            assert got is not None
            np.testing.assert_array_equal(got.cpu().numpy(), want)
            res = kernels.change_detection_pixel_major(dev[0], c12.real, c12.imag, dev[3], alpha=alpha, n=9,
                                                       stats=True)
            np.testing.assert_array_equal(res[0].cpu().numpy(), want)
            np.testing.assert_allclose(res[1].cpu().numpy(), zw, rtol=1e-5, equal_nan=True)
            np.testing.assert_allclose(res[2].cpu().numpy(), pw, rtol=1e-5, atol=1e-7, equal_nan=True)
        # low thresholds: declined (the planar streaming search behind a transpose is the form for them)
        assert kernels.change_detection_pixel_major(dev[0], dev[1], dev[2], dev[3], alpha=0.01, n=9) is None
    t = torch.ones((4, 5, 30), device=device)                     # 30 dates: not whole 16-byte vectors
    assert kernels.change_detection_pixel_major(t, t, t, t, alpha=0.9) is None
    t = torch.ones((2, 3, 196), device=device)
    assert kernels.change_detection_pixel_major(t, t, t, t, alpha=0.9) is None


@pytest.mark.parametrize('dtype,k', [(np.float32, 33), (np.float32, 40), (np.float32, 48), (np.float32, 57),
                                     (np.float32, 64), (np.float64, 17), (np.float64, 40), (np.float64, 64),
                                     (np.float32, 65), (np.float32, 96), (np.float32, 127), (np.float32, 128),
                                     (np.float64, 100), (np.float32, 129)])
def test_long_series_at_low_thresholds(oracle, device, dtype, k):
    """33 .. 128 dates (float64: 17 .. 128) at the thresholds users pass: the streaming search with
    64- / 128-bit masks (`stream_long`), planar and strided inputs, against the oracle byte for
    byte (129 dates: beyond the fast forms, still exact)."""
    import torch
    from nd_amd import kernels
    planes = synth.omnibus_stack(seed=300 + k, k=k, ny=24 if k <= 64 else 6, nx=200, dtype=dtype, change_frac=0.3)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    dev_tyx = [torch.from_numpy(p).to(device) for p in planes]
    dev_yxt = [torch.from_numpy(p).to(device) for p in yxt]
    for alpha in (1e-4, 0.01, 0.2):
        want = oracle.change_detection_planes(yxt, alpha, 9, njobs=8)
        got = kernels.change_detection(*dev_tyx, alpha=alpha, n=9, dims=('time', 'y', 'x'))
        got2 = kernels.change_detection(*dev_yxt, alpha=alpha, n=9, dims=('y', 'x', 'time'))
        torch.cuda.synchronize()
        assert int((got.cpu().numpy() != want).sum()) == 0, (k, alpha)
        assert int((got2.cpu().numpy() != want).sum()) == 0, (k, alpha, 'strided')
        assert want.sum() > 0
    # with the z / P rasters on top (they come from a separate launch of the plain pass A)
    want = oracle.change_detection_planes(yxt, 0.01, 9, njobs=8, stats=True)
    got = kernels.change_detection(*dev_tyx, alpha=0.01, n=9, dims=('time', 'y', 'x'), stats=True)
    torch.cuda.synchronize()
    _compare(tuple(g.cpu().numpy() for g in got), want)


@pytest.mark.parametrize('dtype,k', [(np.float32, 24), (np.float32, 40), (np.float64, 12), (np.float32, 80)])
def test_nodata_margins_at_low_thresholds(oracle, device, dtype, k):
    """Nodata as real products carry it -- NaN or zero fill over whole columns, a NaN / zero / inf
    at a single date -- at the thresholds that run the streaming search, which ends such pixels at
    once (no change anywhere, nd/_change.pyx:239-242) instead of handing them to the exact pass."""
    import torch
    from nd_amd import kernels
    planes = [p.copy() for p in synth.omnibus_stack(seed=500 + k, k=k, ny=16, nx=260, dtype=dtype, change_frac=0.3)]
    for p in planes:
        p[:, :, 0:40] = np.nan                     # NaN margin, every date
        p[:, :, 40:80] = 0.0                       # zero margin, every date
    planes[0][3, :, 80:110] = np.nan               # one variable, one date
    planes[2][k - 1, :, 110:140] = np.nan
    for p in planes:
        p[1, :, 140:170] = 0.0                     # a zero matrix at one date
    planes[3][0, :, 170:200] = np.inf
    planes[1][2, :, 200:215] = -np.inf
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    dev = [torch.from_numpy(p).to(device) for p in planes]
    for alpha in (1e-4, 0.01, 0.3, 0.99):
        with np.errstate(all='ignore'):
            want = oracle.change_detection_planes(yxt, alpha, 9, njobs=8)
        got = kernels.change_detection(*dev, alpha=alpha, n=9, dims=('time', 'y', 'x'))
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        assert int((got != want).sum()) == 0, (k, alpha)
        assert not got[:, 0:215].any()             # nodata never changes
        assert got[:, 215:].any()


@pytest.mark.parametrize('dtype,k', [(np.float32, k_) for k_ in (49, 57, 64, 80, 96, 97, 130, 192)] +
                         [(np.float64, k_) for k_ in (25, 33, 48, 49, 64, 80, 96, 97)])
def test_long_series_sparse_regime(oracle, device, dtype, k):
    """49 .. 192 float32 dates (25 .. 96 float64 dates; 97: beyond, the plain pass A) in the sparse regime
    (alpha >= 0.75): the time-split pass A
    (omnibus_c2_split_kernel: four or eight waves share a pixel's time axis, candidates dumped from
    registers) and the LDS-DMA ring search behind it; with the minimal workspace (no dump: every
    candidate gathered from the planes) and with degenerate values -- whatever the re-associated screen
    cannot vouch for must reach the exact pass -- the map equals the oracle's byte for byte."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(k)
    planes = [p.copy() for p in synth.omnibus_stack(seed=700 + k, k=k, ny=5, nx=333, dtype=dtype,
                                                    change_frac=0.25)]
    planes[1][:, 0, 10:40] *= 6.0                      # |C12|^2 > C11 C22: not positive semi-definite
    planes[0][:, 1, 5:25] *= -1.0
    for p in planes:
        p[:, 2, 0:60] *= 1e-12                         # the product of determinants underflows
        p[:, 2, 60:120] *= 1e10                        # ... overflows
        p[k // 2:, 3, 0:50] *= 1e-9                    # prefix product in range, suffix far below
        p[:, 3, 100:130] = 0.0                         # nodata
        p[:, 3, 130:160] = np.nan
    planes[3][k - 1, 4, 0:30] = np.inf
    planes[2][0, 4, 30:60] = np.nan
    for val in (0.0, -1.0):
        m = rng.random(planes[0].shape) < 0.002
        planes[int(rng.integers(0, 4))][m] = val
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    dev = [torch.from_numpy(p).to(device) for p in planes]
    for alpha in (0.8, 0.99):
        with np.errstate(all='ignore'):
            want = oracle.change_detection_planes(yxt, alpha, 9, njobs=8)
        for ws in ('recommended', 'minimal'):
            got = kernels.change_detection(*dev, alpha=alpha, n=9, dims=('time', 'y', 'x'), workspace=ws)
            torch.cuda.synchronize()
            assert int((got.cpu().numpy() != want).sum()) == 0, (k, alpha, ws)
        assert want.sum() > 0
