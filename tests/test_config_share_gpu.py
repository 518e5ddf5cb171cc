"""Parity at one GPU's share of BASELINE.json configs 4 and 5 (the sizes an 8-GPU run gives every
rank), which cross the 2^31-byte plane-offset limit of the fast omnibus path and exercise the
32-bit pixel lists at 33.5 M pixels:

  config 4  OmnibusTest full-pol C3, 48 dates x 8192 x 8192 over 8 GPUs -> 48 x 1024 x 8192 x 9 planes
            (14.5 GB): ~100 k sampled pixels + the first / last rows against the generic-p oracle
            (mirrors tests/test_fullsize_gpu.py::test_sampled_pixels_and_rows_match_oracle).
  config 5  NLMeansFilter -> OmnibusTest, 24 dates x 16384 x 16384 over 8 GPUs -> 24 x 2048 x 16384 x 4
            (12.9 GB) with the tutorial's parameters (examples/tutorial_s1.ipynb cells 11 and 15:
            nlmeans(dims=('time','y','x'), r=(1,3,3), n_eff=50) with the defaults sigma=1, h=1, f=1,
            then change_omnibus(n=50, alpha=1e-4)): crops at the raster's corners, edges and middle,
            each with its halo, against the oracle's filter and the oracle's test of the oracle's
            filtered values.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_c3_share_sampled_against_oracle(oracle, device):
    import torch
    from nd_amd import kernels, synth
    from oracle import checks
    k, ny, nx = 48, 1024, 8192
    st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=77, device=device, change_frac=0.01)
    torch.cuda.synchronize()
    planes = [st[c] for c in range(9)]
    for alpha in (0.99, 0.01):
        ch = kernels.change_detection_c3(planes, alpha=alpha, n=9)
        torch.cuda.synchronize()
        res = checks.omnibus_sample(st, ch, alpha, 9, nsample=100000 if alpha == 0.99 else 4000,
                                    rows=(0, ny - 1) if alpha == 0.99 else (), seed=9, pol=3)
        assert res['bad'] == 0, res
        if alpha == 0.99:
            assert 0.003 < res['flagged_fraction'] < 0.08, res
        else:
            assert res['flagged_fraction'] > 0.9, res
    # row chunks of the same stack give the same map (the rows are independent)
    ch = kernels.change_detection_c3(planes, alpha=0.99, n=9)
    part = kernels.change_detection_c3([p[:, 500:700] for p in planes], alpha=0.99, n=9)
    assert torch.equal(part, ch[500:700])


@pytest.fixture(scope='module')
def share5(device):
    import torch
    from nd_amd import synth
    k, ny, nx = 24, 2048, 16384
    st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=55, device=device, change_frac=0.01)
    torch.cuda.synchronize()
    return st


_CROPS5 = [(0, 0), (0, 8000), (1000, 16384), (2048, 16384), (2048, 5000), (1017, 7777)]


def test_config5_pipeline_share_tutorial_parameters(oracle, device, share5):
    """reference-compatible patch distances (patch_mode 0): filter values and change map exact."""
    import torch
    from nd_amd import tiles
    from oracle import checks
    k, ny, nx = 24, 2048, 16384
    r, f, sigma, h, n_eff, n, alpha = (1, 3, 3), (1, 1, 1), 1.0, 1.0, 50.0, 50, 1e-4
    filtered = tiles.nlmeans_rows(share5, ny, r, f, sigma, h, n_eff=n_eff, patch_mode=0)
    change = tiles.omnibus_rows(filtered, alpha, n)
    both = tiles.nlmeans_then_omnibus(share5, ny, r, f, sigma, h, alpha, n, n_eff=n_eff, patch_mode=0)
    torch.cuda.synchronize()
    assert torch.equal(both, change)
    res = checks.nlmeans_crops(share5, filtered, r, f, sigma, h, n_eff, 0, _CROPS5, size=(10, 80),
                               then_omnibus=(alpha, n), change=change)
    assert res['bad'] == 0 and res['compared'] > 0, res
    assert res['change_bad'] == 0 and res['change_compared'] > 0, res
    # the test on the unfiltered stack at this size, sampled (the non-EXACT addressing path)
    raw = tiles.omnibus_rows(share5, 0.99, 9)
    res = checks.omnibus_sample(share5, raw, 0.99, 9, nsample=60000, rows=(0, ny - 1), seed=2)
    assert res['bad'] == 0, res
    # the tutorial's thresholds on the raw stack too: almost every date of every pixel changes
    raw = tiles.omnibus_rows(share5, alpha, 9)
    res = checks.omnibus_sample(share5, raw, alpha, 9, nsample=20000, seed=3)
    assert res['bad'] == 0 and res['flagged_fraction'] > 0.9, res


def test_config5_signed_patch_distances(oracle, device, share5):
    """patch_mode 1 (the patch distances the source text describes) on the same share: filter
    values within 1e-5 of the oracle's double arithmetic."""
    import torch
    from nd_amd import tiles
    from oracle import checks
    ny = 2048
    r, f = (1, 3, 3), (1, 1, 1)
    # n_eff = 50 has no solution at pixels whose neighbourhood weights are all small (a x4 step next
    # to stationary pixels): the reference raises ValueError('No solution') there, and so do we
    with pytest.raises(ValueError, match='No solution'):
        tiles.nlmeans_rows(share5, ny, r, f, 1.0, 1.0, n_eff=50.0, patch_mode=1)
    filtered = tiles.nlmeans_rows(share5, ny, r, f, 1.0, 1.0, n_eff=-1, patch_mode=1)
    torch.cuda.synchronize()
    res = checks.nlmeans_crops(share5, filtered, r, f, 1.0, 1.0, -1, 1, _CROPS5[:4], size=(8, 64))
    assert res['bad'] == 0, res
