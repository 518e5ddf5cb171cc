"""GPU parity of the non-local-means HIP path (through the C ABI) against the REAL reference
kernel (golden vectors from oracle/_ref) and the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'nlmeans_ref.npz')
RTOL = 1e-5      # north_star tolerance for floats; most cases are bit-exact


def _gpu_nlm(a, r, f, sigma, h, n_eff, device, patch_mode=0, neff_policy=1, permute=None):
    import torch
    from nd_amd import kernels
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if permute is not None:
        # same logical array, different memory order (planar [var][...])
        inv = np.argsort(permute)
        t = t.permute(*permute).contiguous().permute(*inv)
    out = torch.empty_like(t)
    kernels.pixelwise_nlmeans_3d(t, out, r, f, sigma, h, n_eff, patch_mode=patch_mode,
                                 neff_policy=neff_policy)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _cases():
    g = np.load(GOLD)
    return sorted({n.split('__')[0] for n in g.files if '__' in n})


@pytest.mark.parametrize('name', _cases())
@pytest.mark.parametrize('permute', [None, (3, 2, 0, 1)])
def test_golden_reference(device, name, permute):
    g = np.load(GOLD)
    a, par, want = g[name + '__in'], g[name + '__par'], g[name + '__out']
    r, f = par[:3].astype(int), par[3:6].astype(int)
    got = _gpu_nlm(a, r, f, par[6], par[7], par[8], device, permute=permute)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)
    if (f > 0).any() and par[8] < 0:
        # compiled-reference semantics: all weights are exactly 1 -> same f32 sums, bit for bit
        np.testing.assert_array_equal(got, want)


def test_patch_mode_1_pinned_through_reference(device):
    g = np.load(GOLD)
    a, par, want = g['pm1_in'], g['pm1_par'], g['pm1_out_interior']
    r, f = par[:3].astype(int), par[3:6].astype(int)
    got = _gpu_nlm(a[:, :, None, None], r, f, par[6], par[7], par[8], device, patch_mode=1)
    m, n = r[0] + f[0], r[1] + f[1]
    np.testing.assert_allclose(got[m:-m, n:-n, 0, 0], want, rtol=RTOL, atol=0)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('patch_mode', [0, 1])
def test_oracle_random(oracle, device, dtype, patch_mode):
    rng = np.random.default_rng(12)
    for shape, r, f, s, h, ne in [((13, 17, 3, 2), (2, 3, 1), (1, 1, 1), 0.5, 0.7, -1),
                                  ((24, 25, 1, 1), (4, 4, 0), (2, 2, 0), 0.3, 0.4, -1),
                                  ((9, 8, 4, 3), (1, 2, 1), (0, 0, 0), 0.2, 0.6, -1),
                                  ((10, 11, 2, 4), (2, 2, 0), (1, 1, 0), 1.0, 3.0, 4.0),
                                  ((1, 16, 16, 2), (0, 2, 2), (0, 1, 1), 0.5, 0.5, -1)]:
        a = rng.gamma(4.0, 0.25, shape).astype(dtype)
        want = np.empty_like(a)
        oracle.pixelwise_nlmeans_3d(a, want, r, f, s, h, ne, neff_policy=0, njobs=8,
                                    patch_mode=patch_mode)
        got = _gpu_nlm(a, r, f, s, h, ne, device, patch_mode=patch_mode, neff_policy=0)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)


def test_zero_radius_is_identity(device):
    """nd/tests/test_nlmeans_filter.py:17-25."""
    rng = np.random.default_rng(1)
    a = rng.normal(size=(20, 20, 10, 4))
    got = _gpu_nlm(a, (0, 0, 0), (0, 0, 0), 1, 1, -1, device)
    np.testing.assert_array_equal(got, a)


def test_no_solution_raises(device):
    """find_weight fails where n_eff - 1 > W^2 / W2 (nd/_filters.pyx:310-311): ValueError like a
    current build of the reference; policy 0 gives the self weight 0 of the shipped C."""
    rng = np.random.default_rng(2)
    a = rng.normal(size=(12, 12, 1, 1)).astype(np.float32)
    with pytest.raises(ValueError, match='No solution'):
        _gpu_nlm(a, (2, 2, 0), (0, 0, 0), 0.1, 0.1, 30.0, device)
    out = _gpu_nlm(a, (2, 2, 0), (0, 0, 0), 0.1, 0.1, 30.0, device, neff_policy=0)
    assert out.shape == a.shape


def test_tile_with_halo_equals_untiled(device):
    """y-tiles that carry r+f halo rows reproduce the untiled result (the multi-GPU layout)."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(6)
    a = rng.gamma(4.0, 0.25, (40, 21, 2, 1)).astype(np.float32)
    r, f = (3, 2, 0), (1, 1, 0)
    full = _gpu_nlm(a, r, f, 0.4, 0.5, -1, device, patch_mode=1)
    halo = r[0] + f[0]
    t = torch.from_numpy(a).to(device)
    out = torch.zeros_like(t)
    for lo, hi in [(0, 14), (14, 27), (27, 40)]:
        tlo, thi = max(lo - halo, 0), min(hi + halo, 40)
        tile = t[tlo:thi].contiguous()
        tout = torch.empty_like(tile)
        kernels.pixelwise_nlmeans_3d(tile, tout, r, f, 0.4, 0.5, -1, patch_mode=1,
                                     global_shape=(40, 21, 2), tile_offset=(tlo, 0, 0),
                                     core=((lo - tlo, hi - tlo), (0, 21), (0, 2)))
        out[lo:hi] = tout[lo - tlo:hi - tlo]
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), full)


@pytest.mark.parametrize('patch_mode', [0, 1])
@pytest.mark.parametrize('case', [
    ((70, 131, 2, 1), (3, 4, 0), (1, 1, 0), 0.5, 0.5, -1),
    ((65, 64, 1, 1), (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1),
    ((33, 40, 3, 2), (2, 3, 0), (2, 2, 0), 0.4, 0.6, -1),
    ((40, 70, 2, 4), (3, 2, 0), (1, 1, 0), 0.5, 0.5, -1),
    ((64, 64, 1, 3), (1, 1, 0), (0, 0, 0), 0.3, 0.4, -1),
    ((50, 66, 2, 1), (4, 4, 0), (1, 1, 0), 0.8, 2.0, 6.0),
    ((30, 30, 1, 1), (0, 5, 0), (0, 2, 0), 0.5, 0.5, -1),
])
def test_tiled_kernels_planar_layout(oracle, device, patch_mode, case):
    """Planar [var][time][y][x] memory viewed as (y, x, time, var): the LDS-tiled kernels
    (uniform-weight window sums in patch_mode 0, sliding patch sums in patch_mode 1)."""
    shape, r, f, s, h, ne = case
    rng = np.random.default_rng(41)
    a = rng.gamma(4.0, 0.25, shape).astype(np.float32)
    want = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, want, r, f, s, h, ne, neff_policy=0, njobs=8, patch_mode=patch_mode)
    got = _gpu_nlm(a, r, f, s, h, ne, device, patch_mode=patch_mode, neff_policy=0,
                   permute=(3, 2, 0, 1))
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)
    if patch_mode == 0 and max(f) > 0 and ne < 0:
        np.testing.assert_array_equal(got, want)        # unit weights: bit-exact sums


def test_tiled_halo_tile_equals_untiled(device):
    """Tiled kernels with global_shape / tile_offset / core (multi-GPU row blocks)."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(43)
    full_np = rng.gamma(4.0, 0.25, (1, 2, 96, 70)).astype(np.float32)     # (var, t, y, x)
    t = torch.from_numpy(full_np).to(device)
    r, f = (4, 3, 0), (2, 2, 0)
    for pm in (0, 1):
        ref = torch.empty_like(t)
        kernels.pixelwise_nlmeans_3d(t.permute(2, 3, 1, 0), ref.permute(2, 3, 1, 0), r, f, 0.4, 0.5,
                                     -1, patch_mode=pm)
        halo = r[0] + f[0]
        out = torch.zeros_like(t)
        for lo, hi in [(0, 30), (30, 66), (66, 96)]:
            tlo, thi = max(lo - halo, 0), min(hi + halo, 96)
            tile = t[:, :, tlo:thi].contiguous()
            tout = torch.empty_like(tile)
            kernels.pixelwise_nlmeans_3d(tile.permute(2, 3, 1, 0), tout.permute(2, 3, 1, 0), r, f,
                                         0.4, 0.5, -1, patch_mode=pm, global_shape=(96, 70, 2),
                                         tile_offset=(tlo, 0, 0),
                                         core=((lo - tlo, hi - tlo), (0, 70), (0, 2)))
            out[:, :, lo:hi] = tout[:, :, lo - tlo:hi - tlo]
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_tiny_weights_follow_the_reference(oracle, device):
    """h far too small: every neighbour weight underflows float32 but not double.  The reference
    then still uses max(weight) (a denormal-scale double) as self weight and its float32 weighted
    sums underflow to 0; the tiled kernel must take the same path, not the 'all weights zero' one."""
    rng = np.random.default_rng(44)
    a = rng.gamma(4.0, 0.25, (40, 70, 1, 1)).astype(np.float32)
    for s, h in [(0.1, 0.05), (0.05, 0.02)]:
        want = np.empty_like(a)
        oracle.pixelwise_nlmeans_3d(a, want, (3, 3, 0), (1, 1, 0), s, h, -1, njobs=8, patch_mode=1)
        got = _gpu_nlm(a, (3, 3, 0), (1, 1, 0), s, h, -1, device, patch_mode=1, permute=(3, 2, 0, 1))
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize('case', [
    ((5, 40, 70, 4), (1, 3, 3), (1, 1, 1), 0.5, 0.5, 50.0),      # the tutorial's filter
    ((5, 40, 70, 4), (1, 3, 3), (1, 1, 1), 0.5, 0.5, -1),
    ((3, 33, 130, 1), (2, 2, 4), (1, 0, 1), 0.5, 0.5, -1),
    ((6, 20, 20, 2), (0, 3, 2), (0, 1, 1), 0.5, 0.5, -1),         # time is a plain slice axis
    ((4, 35, 66, 2), (1, 0, 2), (1, 0, 1), 0.4, 0.6, 20.0),
    # three-date windows of every size the streaming kernel is instantiated for, series of 2 .. 7 dates
    ((2, 37, 131, 1), (1, 1, 1), (1, 1, 1), 0.5, 0.5, -1),
    ((3, 37, 131, 2), (1, 2, 2), (1, 1, 1), 0.5, 0.5, 9.0),
    ((7, 70, 259, 1), (1, 4, 4), (1, 1, 1), 0.5, 0.5, -1),
    ((4, 33, 140, 2), (1, 5, 5), (1, 2, 2), 0.5, 0.5, 30.0),
    # five- and seven-date windows: the ring kernel with more than 64 KB of LDS
    ((7, 40, 150, 2), (2, 3, 3), (1, 1, 1), 0.5, 0.5, -1),
    ((9, 36, 131, 1), (3, 2, 2), (1, 1, 1), 0.5, 0.5, 12.0),
])
def test_time_first_layout_window_kernel(oracle, device, case):
    """(time, y, x, var) views of planar stacks, 3-D search window: the reference-compatible mode
    (unit weights) runs in the tiled window kernel with the time offset visited outermost, as in
    nd/_filters.pyx:363-370; results must equal the reference's float32 sums bit for bit."""
    import torch
    from nd_amd import kernels
    shape, r, f, s, h, ne = case
    rng = np.random.default_rng(47)
    a = rng.gamma(4.0, 0.25, shape).astype(np.float32)                 # (t, y, x, var)
    want = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, want, r, f, s, h, ne, neff_policy=0, njobs=8, patch_mode=0)
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)   # (var, t, y, x)
    out = torch.empty_like(planar)
    kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, s, h, ne,
                                 patch_mode=0, neff_policy=0)
    torch.cuda.synchronize()
    got = out.permute(1, 2, 3, 0).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_time_first_layout_patch_kernel_and_tiles(oracle, device):
    """(time, y, x, var) views with r_time = 0: true patch distances through the sliding-sum kernel,
    and row blocks with halos along y (user axis 1) reproduce the unsharded result."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(48)
    a = rng.gamma(4.0, 0.25, (3, 70, 66, 2)).astype(np.float32)
    r, f = (0, 3, 2), (0, 1, 1)
    want = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, want, r, f, 0.4, 0.5, -1, njobs=8, patch_mode=1)
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
    out = torch.empty_like(planar)
    kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, 0.4, 0.5, -1,
                                 patch_mode=1)
    np.testing.assert_allclose(out.permute(1, 2, 3, 0).cpu().numpy(), want, rtol=RTOL)
    for pm, rr, ff in ((1, r, f), (0, (1, 3, 2), (1, 1, 1))):
        ref = torch.empty_like(planar)
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), ref.permute(1, 2, 3, 0), rr, ff, 0.4, 0.5,
                                     -1, patch_mode=pm)
        halo = rr[1] + ff[1]
        acc = torch.zeros_like(planar)
        for lo, hi in [(0, 25), (25, 48), (48, 70)]:
            tlo, thi = max(lo - halo, 0), min(hi + halo, 70)
            tile = planar[:, :, tlo:thi].contiguous()
            tout = torch.empty_like(tile)
            kernels.pixelwise_nlmeans_3d(tile.permute(1, 2, 3, 0), tout.permute(1, 2, 3, 0), rr, ff, 0.4,
                                         0.5, -1, patch_mode=pm, global_shape=(3, 70, 66),
                                         tile_offset=(0, tlo, 0),
                                         core=((0, 3), (lo - tlo, hi - tlo), (0, 66)))
            acc[:, :, lo:hi] = tout[:, :, lo - tlo:hi - tlo]
        torch.cuda.synchronize()
        assert torch.equal(acc, ref)


def test_window_kernel_magnitudes_and_rounding(oracle, device):
    """Unit-weight window kernel over data spanning the float32 range (zeros, negatives, values near
    the subnormal and overflow ends): the normalisation (float)((double)sum / total) must round like
    the reference's division for every element, with integer and fractional totals."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(52)
    shape = (4, 96, 260, 2)
    a = rng.gamma(4.0, 0.25, shape)
    scale = 10.0 ** rng.integers(-36, 34, (1, 96 // 8, 260 // 10, 1)).astype(np.float64)
    a = a * np.repeat(np.repeat(scale, 8, axis=1), 10, axis=2)
    a[:, 10:20, 30:80] = 0.0
    a[:, 40:60, 100:150] *= -1.0
    a = a.astype(np.float32)
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
    for r, f, ne in (((1, 3, 3), (1, 1, 1), 50.0), ((1, 3, 3), (1, 1, 1), -1), ((0, 10, 10), (0, 3, 3), -1),
                     ((1, 2, 5), (1, 1, 1), 7.5)):
        want = np.empty_like(a)
        with np.errstate(all='ignore'):
            oracle.pixelwise_nlmeans_3d(a, want, r, f, 0.5, 0.5, ne, neff_policy=0, njobs=8, patch_mode=0)
        out = torch.empty_like(planar)
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne,
                                     patch_mode=0, neff_policy=0)
        torch.cuda.synchronize()
        got = out.permute(1, 2, 3, 0).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_window_kernel_partial_core_on_every_axis(device):
    """Written range restricted along the window's outermost axis too (a tile of a longer series)."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(53)
    full = torch.from_numpy(rng.gamma(4.0, 0.25, (2, 9, 50, 140)).astype(np.float32)).to(device)  # (var,t,y,x)
    _partial_core_case(full, (2, 3, 3), (1, 1, 1))
    _partial_core_case(full, (1, 3, 3), (1, 1, 1))          # three-date window: the streaming kernel


def _partial_core_case(full, r, f):
    import torch
    from nd_amd import kernels
    ref = torch.empty_like(full)
    kernels.pixelwise_nlmeans_3d(full.permute(1, 2, 3, 0), ref.permute(1, 2, 3, 0), r, f, 0.5, 0.5, -1,
                                 patch_mode=0)
    acc = torch.zeros_like(full)
    for lo, hi in [(0, 4), (4, 9)]:
        tlo, thi = max(lo - r[0], 0), min(hi + r[0], 9)
        tile = full[:, tlo:thi].contiguous()
        tout = torch.full_like(tile, -7.0)
        kernels.pixelwise_nlmeans_3d(tile.permute(1, 2, 3, 0), tout.permute(1, 2, 3, 0), r, f, 0.5, 0.5, -1,
                                     patch_mode=0, global_shape=(9, 50, 140), tile_offset=(tlo, 0, 0),
                                     core=((lo - tlo, hi - tlo), (3, 47), (5, 133)))
        acc[:, lo:hi, 3:47, 5:133] = tout[:, lo - tlo:hi - tlo, 3:47, 5:133]
        # nothing outside the core is written
        probe = tout.clone()
        probe[:, lo - tlo:hi - tlo, 3:47, 5:133] = -7.0
        assert bool((probe == -7.0).all())
    torch.cuda.synchronize()
    assert torch.equal(acc[:, :, 3:47, 5:133], ref[:, :, 3:47, 5:133])


def test_window_kernels_with_nodata(oracle, device):
    """NaN / inf margins and isolated NaNs under the unit-weight window kernels (streaming
    three-date form, ring form, 2-D form) and the signed patch kernels: NaN spreads over exactly
    the windows that hold it, as in the reference."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(91)
    a = rng.gamma(4.0, 0.25, (5, 40, 150, 2)).astype(np.float32)          # (t, y, x, var)
    a[:, :, 0:12, :] = np.nan
    a[2, 20, 70, 0] = np.nan
    a[4, 5, 100, 1] = np.inf
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
    for r, f, ne, pm in (((1, 3, 3), (1, 1, 1), 50.0, 0), ((2, 2, 2), (1, 1, 1), -1, 0), ((0, 3, 3), (0, 1, 1), -1, 0),
                         ((0, 3, 3), (0, 1, 1), -1, 1), ((0, 2, 2), (0, 1, 1), 6.0, 1),
                         # f = 0: the one-column-per-lane patch kernel, whose sliding sum down the
                         # column must not carry a NaN / inf row sum along
                         ((0, 3, 3), (0, 0, 0), -1, 1), ((0, 2, 2), (0, 0, 0), 6.0, 1)):
        want = np.empty_like(a)
        with np.errstate(all='ignore'):
            oracle.pixelwise_nlmeans_3d(a, want, r, f, 0.5, 0.5, ne, neff_policy=0, njobs=8, patch_mode=pm)
        out = torch.empty_like(planar)
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne,
                                     patch_mode=pm, neff_policy=0)
        torch.cuda.synchronize()
        got = out.permute(1, 2, 3, 0).cpu().numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=str((r, f, ne, pm)))
        if pm == 0:
            np.testing.assert_array_equal(got, want)
        else:
            ok = ~np.isnan(want)
            np.testing.assert_allclose(got[ok], want[ok], rtol=RTOL)


def test_signed_mode_patches_of_9x9_and_11x11(oracle, device):
    """f = 4 and 5 in the signed mode run in the one-column-per-lane patch kernel (round 6; the per-pixel kernel
    before: 500 x slower than f = 2): 1 - 4 variables, n_eff, ragged tiles, NaN / inf nodata -- 1e-5 against the
    oracle's double arithmetic."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(451)
    for nv, shape in ((1, (2, 70, 131)), (3, (1, 37, 70)), (4, (2, 20, 65))):
        a = rng.gamma(4.0, 0.25, shape + (nv,)).astype(np.float32)          # (t, y, x, var)
        planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
        for r, f, ne in (((0, 5, 5), (0, 4, 4), -1), ((0, 6, 6), (0, 5, 5), -1), ((0, 4, 4), (0, 4, 4), 30.0)):
            want = np.empty_like(a)
            oracle.pixelwise_nlmeans_3d(a, want, r, f, 0.5, 0.5, ne, neff_policy=0, njobs=8, patch_mode=1)
            out = torch.empty_like(planar)
            kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne,
                                         patch_mode=1, neff_policy=0)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.permute(1, 2, 3, 0).cpu().numpy(), want, rtol=RTOL, err_msg=str((nv, r, f, ne)))
    a = rng.gamma(4.0, 0.25, (1, 40, 90, 2)).astype(np.float32)
    a[0, 10, 20, 0] = np.nan
    a[0, 30, 60, 1] = np.inf
    a[0, :, :6, :] = np.nan
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
    want = np.empty_like(a)
    with np.errstate(all='ignore'):
        oracle.pixelwise_nlmeans_3d(a, want, (0, 5, 5), (0, 4, 4), 0.5, 0.5, -1, neff_policy=0, njobs=8, patch_mode=1)
    out = torch.empty_like(planar)
    kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), (0, 5, 5), (0, 4, 4), 0.5, 0.5, -1,
                                 patch_mode=1, neff_policy=0)
    torch.cuda.synchronize()
    got = out.permute(1, 2, 3, 0).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=RTOL)


def test_signed_mode_far_on_the_no_solution_side(oracle, device):
    """n_eff beyond what the neighbours can give (n_eff - 1 > W^2 / W2 at every pixel: find_weight raises at each,
    the reference's self weight is 0 under neff_policy 0).  The tiled kernels keep those pixels in the fast path
    when the verdict is more than 5 % clear (round 6; before, every pixel was recomputed one by one) and send the
    neighbourhood of the boundary to the exact path: n_eff = 60 / 26 with 48 neighbours covers both, as do the
    3-D search (146 neighbours) and the f = 0 kernel."""
    import torch
    from nd_amd import kernels
    rng = np.random.default_rng(191)
    a = rng.gamma(4.0, 0.25, (4, 37, 141, 2)).astype(np.float32)          # (t, y, x, var)
    planar = torch.from_numpy(np.ascontiguousarray(a.transpose(3, 0, 1, 2))).to(device)
    for r, f, ne in (((0, 3, 3), (0, 1, 1), 60.0), ((0, 3, 3), (0, 1, 1), 26.0), ((0, 3, 3), (0, 0, 0), 60.0),
                     ((1, 3, 3), (1, 1, 1), 200.0), ((1, 3, 3), (0, 1, 1), 75.0)):
        want = np.empty_like(a)
        oracle.pixelwise_nlmeans_3d(a, want, r, f, 0.5, 0.5, ne, neff_policy=0, njobs=8, patch_mode=1)
        out = torch.empty_like(planar)
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne,
                                     patch_mode=1, neff_policy=0)
        torch.cuda.synchronize()
        got = out.permute(1, 2, 3, 0).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=str((r, f, ne)))
    # neff_policy 1: the reference's ValueError
    with pytest.raises(ValueError, match='No solution'):
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), (0, 3, 3), (0, 1, 1), 0.5, 0.5,
                                     60.0, patch_mode=1, neff_policy=1)


def test_one_column_patch_kernel_forced(tmp_path):
    """The one-column-per-lane patch kernel normally serves only f = 0 and tiles beyond 150 KB of
    LDS; forced for every signed-mode case (ND_AMD_NLM_PATCH1, read once per process, hence the one
    child process) it must pass the same nodata, random-oracle and tiling-invariance checks -- its
    sliding sum down the column once carried NaN row sums along."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, ND_AMD_NLM_PATCH1='1')
    with open(tmp_path / 'log', 'w') as fo:
        p = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider',
                            '-k', 'nodata or oracle_random or halo or tiny_weights or planar_layout'],
                           env=env, stdout=fo, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=900,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    log = (tmp_path / 'log').read_text()
    assert p.returncode == 0, log[-3000:]
    assert ' passed' in log and 'failed' not in log, log[-1000:]


@pytest.mark.parametrize('case', [
    # (nvars, k, ny, nx, r, f, sigma, h, n_eff)
    (4, 6, 45, 150, (1, 3, 3), (1, 1, 1), 0.5, 0.5, -1),       # the tutorial's window, 4 variables
    (4, 5, 40, 131, (1, 3, 3), (1, 1, 1), 0.5, 0.5, 30.0),     # ... with n_eff (find_weight)
    (1, 7, 37, 200, (2, 4, 4), (1, 1, 1), 0.3, 0.4, -1),       # two dates either side, radius 4
    (2, 4, 33, 90, (1, 2, 2), (0, 1, 1), 0.4, 0.5, -1),        # no patch extent along time
    (3, 5, 50, 70, (1, 3, 3), (1, 2, 2), 0.5, 0.6, -1),        # 5 x 5 x 3 patches: the per-pixel kernel (not tiled)
    (4, 7, 30, 100, (2, 3, 3), (1, 1, 1), 0.5, 0.5, -1),       # two dates either side, 4 variables: 8-row tiles
    (3, 6, 26, 90, (2, 2, 2), (0, 1, 1), 0.4, 0.5, 20.0),      # ... 3 variables, n_eff
    (2, 4, 20, 64, (1, 1, 1), (1, 0, 0), 0.4, 0.5, -1),        # patch along time only
])
def test_signed_mode_search_along_time_tiled_kernel(oracle, device, case):
    """patch_mode 1 with a search (and patch) extent along time on (time, y, x) arrays: nlmeans_patch3_kernel
    (staged planes, the thread's own patch values in registers, cross-lane row sums) against the oracle's
    double arithmetic, 1e-5 relative -- ragged tiles, several variables, n_eff, NaN / inf / zero nodata, and a
    row tile with halo in global coordinates (nd/_filters.pyx:363-420)."""
    import torch
    from nd_amd import kernels
    nv, k, ny, nx, r, f, sigma, h, ne = case
    rng = np.random.default_rng(nv * 100 + k)
    a = rng.gamma(4.0, 0.25, size=(k, ny, nx, nv)).astype(np.float32)
    a[2, 5:9, 20:30, :] *= 6.0                                  # a bright block: tiny weights around it
    for label, arr in (('plain', a), ('nodata', None)):
        if arr is None:
            arr = a.copy()
            arr[:, 0:6, 0:40, :] = np.nan                       # a NaN margin
            arr[1, min(20, ny - 1), 50, 0] = np.inf
            arr[:, 30:33, 100:120, :] = 0.0
        want = np.empty_like(arr)
        with np.errstate(all='ignore'):
            oracle.pixelwise_nlmeans_3d(arr, want, r, f, sigma, h, ne, neff_policy=0, njobs=8, patch_mode=1)
        planar = torch.from_numpy(np.ascontiguousarray(np.transpose(arr, (3, 0, 1, 2)))).to(device)   # (v, t, y, x)
        out = torch.empty_like(planar)
        kernels.pixelwise_nlmeans_3d(planar.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), r, f, sigma, h, ne,
                                     patch_mode=1, neff_policy=0)
        torch.cuda.synchronize()
        got = np.transpose(out.cpu().numpy(), (1, 2, 3, 0))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-30, equal_nan=True, err_msg=label)
    # a row tile with its halo, reflection in GLOBAL coordinates: rows [c0, c1) of the raster
    halo = r[1] + f[1]
    c0, c1 = ny // 3, min(ny // 3 + 18, ny)
    lo, hi = max(c0 - halo, 0), min(c1 + halo, ny)
    tile = torch.from_numpy(np.ascontiguousarray(np.transpose(a[:, lo:hi], (3, 0, 1, 2)))).to(device)
    tout = torch.zeros_like(tile)
    kernels.pixelwise_nlmeans_3d(tile.permute(1, 2, 3, 0), tout.permute(1, 2, 3, 0), r, f, sigma, h, ne,
                                 patch_mode=1, neff_policy=0, global_shape=(k, ny, nx), tile_offset=(0, lo, 0),
                                 core=((0, k), (c0 - lo, c1 - lo), (0, nx)))
    torch.cuda.synchronize()
    want = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, want, r, f, sigma, h, ne, neff_policy=0, njobs=8, patch_mode=1)
    got = np.transpose(tout.cpu().numpy(), (1, 2, 3, 0))[:, c0 - lo:c1 - lo]
    np.testing.assert_allclose(got, want[:, c0:c1], rtol=1e-5, atol=1e-30, equal_nan=True)
