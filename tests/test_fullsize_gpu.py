"""Checks at BASELINE.json's full size (24 dates x 4096 x 4096, the benchmark stack itself): the
oracle on a random sample of pixels and on whole rows of the same raster, plus size-independent
properties (row-chunk invariance, a constant series never changes, a strong injected step is found
at its date, the reference-layout view gives the same map)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, NY, NX = 24, 4096, 4096


@pytest.fixture(scope='module')
def stack(device):
    import torch
    from nd_amd import synth
    s = synth.wishart_c2_stack(K, NY, NX, looks=9, seed=1234, device=device, change_frac=0.01)
    torch.cuda.synchronize()
    return s


@pytest.fixture(scope='module')
def full_map(stack):
    import torch
    from nd_amd import kernels
    ch = kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=0.99, n=9)
    torch.cuda.synchronize()
    return ch


def test_sampled_pixels_and_rows_match_oracle(oracle, stack, full_map):
    import torch
    g = torch.Generator(device='cpu').manual_seed(5)
    idx = torch.randint(0, NY * NX, (150000,), generator=g)
    rows = torch.tensor([0, 1, 2047, 4095])
    idx = torch.cat([idx, (rows[:, None] * NX + torch.arange(NX)[None]).reshape(-1)])
    dev_idx = idx.to(stack.device)
    flat = stack.reshape(4, K, NY * NX)
    sample = flat[:, :, dev_idx].cpu().numpy()                     # (4, K, n)
    planes = [np.ascontiguousarray(sample[v].T)[None] for v in range(4)]     # (1, n, K)
    want = oracle.change_detection_planes(planes, 0.99, 9, njobs=8)[0]
    got = full_map.reshape(NY * NX, K)[dev_idx].cpu().numpy()
    assert got.shape == want.shape
    nbad = int((got != want).sum())
    assert nbad == 0, '%d of %d sampled change-map bytes differ' % (nbad, got.size)
    frac = (want.sum(axis=1) > 0).mean()
    assert 0.005 < frac < 0.05                                    # ~1 % injected + ~1 % false alarms


def test_row_chunks_give_the_same_map(stack, full_map):
    import torch
    from nd_amd import kernels
    for r0, r1 in [(0, 1000), (1000, 1001), (3000, 4096)]:
        part = kernels.change_detection(stack[0][:, r0:r1], stack[1][:, r0:r1], stack[2][:, r0:r1],
                                        stack[3][:, r0:r1], alpha=0.99, n=9)
        assert torch.equal(part, full_map[r0:r1])


def test_reference_layout_view_gives_the_same_map(stack, full_map):
    """(y, x, time) strided views of the same memory: the generic-stride path."""
    import torch
    from nd_amd import kernels
    sub = [stack[v][:, 100:164].permute(1, 2, 0) for v in range(4)]           # (y, x, time) views
    got = kernels.change_detection(*sub, alpha=0.99, n=9, dims=('y', 'x', 'time'))
    assert torch.equal(got, full_map[100:164])


def test_constant_series_and_strong_step(device):
    import torch
    from nd_amd import kernels, synth
    k, ny, nx = 24, 512, 4096
    st = synth.empty_stack(4, k, ny, nx, device)
    st[0].fill_(1.0); st[1].fill_(0.1); st[2].fill_(-0.05); st[3].fill_(0.7)
    ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.5, n=9)
    assert int(ch.sum()) == 0                  # identical matrices: z = -0, P = 0
    st[0][13:] *= 50.0
    st[3][13:] *= 50.0
    st[1][13:] *= 50.0
    st[2][13:] *= 50.0
    ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    assert bool(ch[:, :, 13].all())
    assert int(ch.sum()) == ny * nx


def test_large_k_paths(oracle, device):
    """k beyond the register-retaining kernel (streaming + gather) and beyond the by-value table."""
    import torch
    from nd_amd import kernels
    from tests import synth as tsynth
    for k, ny, nx in [(56, 12, 260), (100, 6, 64), (130, 3, 40)]:
        planes = tsynth.omnibus_stack(seed=k, k=k, ny=ny, nx=nx, dtype=np.float32, change_frac=0.3)
        ts = [torch.from_numpy(p).to(device) for p in planes]
        ch, z, P = kernels.change_detection(*ts, alpha=0.9, n=9, stats=True)
        ch2 = kernels.change_detection(*ts, alpha=0.9, n=9)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        want, z0, P0 = oracle.change_detection_planes(yxt, 0.9, 9, njobs=8, stats=True)
        np.testing.assert_array_equal(ch.cpu().numpy(), want)
        np.testing.assert_array_equal(ch2.cpu().numpy(), want)
        np.testing.assert_allclose(z.cpu().numpy(), z0, rtol=1e-5)
        np.testing.assert_allclose(P.cpu().numpy(), P0, rtol=1e-5, atol=1e-30)


def test_nlmeans_config_size_band_matches_oracle(oracle, device):
    """BASELINE config 3 (non-local means 7x7 patch / 21x21 search, 12 dates x 4096 x 4096, one
    band): run the full raster, check a band of it (with its halo) against the oracle in both patch
    modes, and that a constant raster stays constant."""
    import torch
    from nd_amd import kernels
    k, ny, nx = 12, 4096, 4096
    g = torch.Generator(device=device).manual_seed(11)
    x = torch.empty((1, k, ny, nx), device=device)
    for t in range(k):
        u = torch.rand((4, ny, nx), generator=g, device=device)
        x[0, t] = -0.25 * torch.log(u).sum(dim=0)              # Gamma(4, 0.25)
    arr = x.permute(2, 3, 1, 0)                                # (y, x, time, var) view
    r, f, halo = (10, 10, 0), (3, 3, 0), 13
    y0, x0, hh, ww, t_sel = 2040, 3000, 8, 96, 5
    crop = x[:, t_sel:t_sel + 1, y0 - halo:y0 + hh + halo, x0 - halo:x0 + ww + halo]
    crop_np = np.ascontiguousarray(crop.permute(2, 3, 1, 0).cpu().numpy())
    for pm in (0, 1):
        out = torch.empty_like(x)
        kernels.pixelwise_nlmeans_3d(arr, out.permute(2, 3, 1, 0), r, f, 0.5, 0.5, -1, patch_mode=pm)
        torch.cuda.synchronize()
        want = np.empty_like(crop_np)
        oracle.pixelwise_nlmeans_3d(crop_np, want, r, f, 0.5, 0.5, -1, njobs=8, patch_mode=pm)
        got = out[0, t_sel, y0:y0 + hh, x0:x0 + ww].cpu().numpy()
        np.testing.assert_allclose(got, want[halo:halo + hh, halo:halo + ww, 0, 0], rtol=1e-5)
        if pm == 0:
            np.testing.assert_array_equal(got, want[halo:halo + hh, halo:halo + ww, 0, 0])
    c = torch.full((1, 2, 512, 512), 2.5, device=device)
    o = torch.empty_like(c)
    kernels.pixelwise_nlmeans_3d(c.permute(2, 3, 1, 0), o.permute(2, 3, 1, 0), r, f, 0.5, 0.5, -1,
                                 patch_mode=1)
    assert float((o - 2.5).abs().max()) < 1e-6


def test_filters_config_size_bands_match_scipy(device):
    """Boxcar 3x3 / 5x5, a random 5x5 kernel and a Gaussian (sigma 1) on a 24 x 4096 x 4096 float32
    stack (the multilooking step in front of OmnibusTest at BASELINE config size): bands at the top,
    middle and bottom of a few dates, including the raster's borders, equal scipy bit for bit (the
    tiled kernels walking all 24 planes of the batch)."""
    import scipy.ndimage as snf
    import torch
    from nd_amd import kernels
    k, ny, nx = 24, 4096, 4096
    g = torch.Generator(device=device).manual_seed(3)
    x = torch.rand((k, ny, nx), generator=g, device=device) - 0.25
    out = torch.empty_like(x)
    bands = [(0, 48), (2031, 2079), (4048, 4096)]
    dates = [0, 11, 23]
    rng = np.random.default_rng(8)

    def check(kernel2d=None, sigma=None):
        halo = 12
        for t in dates:
            for (r0, r1) in bands:
                e0, e1 = max(r0 - halo, 0), min(r1 + halo, ny)
                # scipy on the band with `halo` rows of context; rows at the raster's own edges keep
                # the true border, the artificial edges of the band are cut off again
                host = x[t, e0:e1].cpu().numpy()
                if kernel2d is not None:
                    want = snf.convolve(host, kernel2d, mode='reflect')
                else:
                    want = snf.gaussian_filter(host, sigma=sigma, mode='reflect')
                lo = halo if e0 > 0 else 0
                hi = want.shape[0] - (halo if e1 < ny else 0)
                got = out[t, e0 + lo:e0 + hi].cpu().numpy()
                np.testing.assert_array_equal(got, want[lo:hi])

    for kern in (np.ones((3, 3)) / 9.0, np.ones((5, 5)) / 25.0, rng.normal(size=(5, 5))):
        kernels.convolve(x, kern[None], out=out)
        check(kernel2d=kern)
    kernels.gaussian_filter(x, (0.0, 1.0, 1.0), out=out)
    check(sigma=1.0)


@pytest.mark.parametrize('alpha', [1e-4, 0.01, 0.5])
def test_dense_thresholds_whole_raster_equals_oracle(oracle, stack, alpha):
    """The thresholds users pass (the reference's default alpha = 0.01, the tutorial's 1e-4) make
    nearly every pixel change at nearly every date: the fused search kernels (streaming form up to
    alpha = 0.05, register form above) against the oracle on the WHOLE 24 x 4096 x 4096 raster."""
    import torch
    from nd_amd import kernels
    ch = kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=alpha, n=9)
    torch.cuda.synchronize()
    host = stack.cpu().numpy()
    planes = [np.moveaxis(host[v], 0, -1) for v in range(4)]
    want = oracle.change_detection_planes(planes, alpha, 9, njobs=16)
    got = ch.cpu().numpy()
    nbad = int((got != want).sum())
    assert nbad == 0, '%d change-map bytes differ at alpha=%g' % (nbad, alpha)
    assert (want.sum(axis=2) > 0).mean() > 0.4


_FORMS_24 = [{'ND_AMD_FUSED_FORM': '0'}, {'ND_AMD_FUSED_FORM': '2'},
             {'ND_AMD_SEARCH_STARTS': '0'}, {'ND_AMD_PM_STREAM_LDS': '0', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_FUSED_ALPHA': '0'}, {'ND_AMD_PM_FORM': '1'}, {'ND_AMD_PM_DIRECT': '0'},
             {'ND_AMD_GATE': '0'}, {'ND_AMD_GATE': '0', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_PM_STREAM_LDS': '1', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_PM_STREAM_LDS': '1', 'ND_AMD_GATE': '0', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_PM_STREAM_SECTOR': '0', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_PM_STREAM_SECTOR': '1', 'ND_AMD_FUSED_FORM': '0'},
             {'ND_AMD_SEARCH_MODE': '0'}, {'ND_AMD_SEARCH_MODE': '1'},
             {'ND_AMD_SEARCH_MODE': '0', 'ND_AMD_FUSED_ALPHA': '0'}]
# the forms that exist at the other series lengths: 12 dates (every form, reference layout included),
# 48 and 96 dates (planar only: streaming search with 64- / 128-bit masks, the chain search in two
# streaming passes, pass B from its LDS image / from memory / one lane per segment start, no fusion)
_FORMS_12 = [{}, {'ND_AMD_FUSED_FORM': '0', 'ND_AMD_PM_DIRECT': '0'}, {'ND_AMD_FUSED_FORM': '2', 'ND_AMD_PM_FORM': '1'},
             {'ND_AMD_FUSED_ALPHA': '0', 'ND_AMD_SEARCH_MODE': '1'}]
_FORMS_LONG = [{}, {'ND_AMD_FUSED_FORM': '0', 'ND_AMD_SEARCH_STARTS': '0'}, {'ND_AMD_FUSED_FORM': '3', 'ND_AMD_GATE': '0'},
               {'ND_AMD_FUSED_ALPHA': '0', 'ND_AMD_SEARCH_MODE': '1'},
               # pass B's LDS image with 64 / 32 series per wave (default from 48 dates on: 16)
               {'ND_AMD_SEARCH_PXW': '64'}, {'ND_AMD_SEARCH_PXW': '32', 'ND_AMD_FUSED_ALPHA': '0'}]


def _bench_dense_many(jobs, workers=4):
    """jobs: [(label, argv of tools/bench_dense.py, environment additions, number of result lines expected)].
    Every form is forced through the environment, which the library reads once per process: one fresh process
    per job, `workers` of them at a time (the GPU box allows six processes on its card; this one counts).
    The oracle's maps depend on the (seeded) stack and the threshold only, not on the form: the first job runs
    alone and leaves them in a directory the others read (the oracle was most of these tests' time).
    -> {label: result lines}; asserts exit codes and line counts."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    with tempfile.TemporaryDirectory(prefix='nd_amd_want_') as wdir:
        def one(job):
            label, argv, env, nlines = job
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_dense.py')] + argv +
                                 ['--want-dir', wdir], env=e, capture_output=True, text=True, timeout=600,
                                 stdin=subprocess.DEVNULL)
            assert out.returncode == 0, (label, out.stderr[-2000:])
            lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')]
            assert len(lines) == nlines, (label, out.stdout[-2000:])
            return label, lines

        # jobs that share their arguments share their maps: the first of each argument list goes first
        first, rest, seen = [], [], set()
        for job in jobs:
            key = tuple(job[1])
            (rest if key in seen else first).append(job)
            seen.add(key)
        res = {}
        with ThreadPoolExecutor(max_workers=workers) as pool:
            res.update(pool.map(one, first))
            res.update(pool.map(one, rest))
        return res


@pytest.mark.parametrize('k,forms', [(24, _FORMS_24), (12, _FORMS_12), (48, _FORMS_LONG), (96, _FORMS_LONG)])
def test_every_kernel_form_gives_the_same_map(k, forms):
    """The library picks among several forms of the search by alpha, series length and a sample of the
    data (streaming fused, chain fused in registers or in two streaming passes, separate dense kernel,
    gate on / off; LDS-DMA or register-staged pixel-major pass A; pixel-major streaming search from
    memory or from LDS images; pass B from its LDS image / from memory / chain form / one lane per
    segment start).  The choice is about speed only: force each form in a fresh process, at 12, 24, 48
    and 96 dates, and compare with the oracle -- also with every pixel of a low-threshold run going
    through pass B (ND_AMD_FUSED_ALPHA=0).  (Round 4 deleted the forms that lost everywhere: the
    register triangle ND_AMD_FUSED_FORM=1 and the round-based pass B ND_AMD_SEARCH_MODE=2.  Round 6: the
    forms of one series length run four processes at a time -- 32 processes one after the other were
    250 s of the suite.)"""
    long_ = k > 24
    layouts = 'planar' if long_ else 'planar,pm'
    argv = ['--k', str(k), '--ny', '256' if long_ else '512', '--nx', '2048', '--alphas', '1e-4,0.01,0.2,0.6,0.99',
            '--steps', '1', '--cpu-rows', '256' if long_ else '512', '--layouts', layouts]
    jobs = [(repr(sorted(env.items())), argv, env, 5 if long_ else 10) for env in forms]
    if long_:
        # (round 6) the sparse regime of 96 dates: the time-split pass A off / pass B from its LDS image /
        # the double-precision screen of the sweep
        jobs += [(repr(sorted(env.items())), argv, env, 5) for env in
                 ({'ND_AMD_C2_SPLIT': '0'}, {'ND_AMD_SEARCH_FS': '0'}, {'ND_AMD_C2_SPLIT': '1', 'ND_AMD_SEARCH_MODE': '0'})]
    for label, lines in _bench_dense_many(jobs).items():
        for r in lines:
            assert r['bytes_differing'] == 0, (label, r)


_CHAIN_LENGTHS = [(20, 'f64', {}), (24, 'f64', {}), (33, 'f32', {}), (32, 'f32', {}), (12, 'f64', {}),
                  (40, 'f32', {}), (63, 'f32', {}), (96, 'f32', {}), (128, 'f32', {}), (40, 'f64', {}),
                  # (round 6) 129 .. 192 dates: three-word masks, pass B with three starts per lane
                  (129, 'f32', {}), (160, 'f32', {}), (192, 'f32', {}), (150, 'f64', {}),
                  (96, 'f32', {'ND_AMD_FUSED_FORM': '3', 'ND_AMD_SEARCH_STARTS': '0'})]


def test_chain_form_series_lengths():
    """dense_chain beyond the 24 float32 dates of the benchmark: 32 float32 / 16 float64 dates (two
    waves per SIMD), the 64-bit-mask instantiation for 17 .. 24 float64 dates (default between the
    streaming search's thresholds and the sparse regime), and beyond the registers the chain search in
    two streaming passes (33 .. 192 dates: 64-, 128- and 192-bit masks, pending global tests, the per-start
    pass B with two or three starts per lane), each against the oracle.  (The 33 .. 48-date float32 register
    instantiation was deleted in round 4: it spilled and never was the default.)"""
    jobs = []
    for k, dtype, env in _CHAIN_LENGTHS:
        # (the oracle's search is quadratic in the series length: fewer rows for the longest series)
        ny = '128' if k > 128 else ('256' if k > 64 else '512')
        argv = ['--k', str(k), '--dtype', dtype, '--ny', ny, '--nx', '2048',
                '--alphas', '0.05,0.3,0.6', '--steps', '1', '--cpu-rows', '128' if k > 128 else '256', '--layouts', 'planar']
        jobs.append(('%d %s %r' % (k, dtype, env), argv, env, 3))
    for label, lines in _bench_dense_many(jobs).items():
        for r in lines:
            assert r['bytes_differing'] == 0, (label, r)
            assert 'omnibus_c2_fused' in r['kernels_ms'], (label, r)


@pytest.mark.parametrize('dtype,k', [('float32', 40), ('float64', 24), ('float32', 80), ('float64', 40), ('float32', 112),
                                      ('float32', 160)])
def test_long_series_statistics_at_scale(oracle, dtype, k):
    """Series beyond the register forms on a raster large enough for the device-side density gate:
    the map of the streaming search (64- / 128-bit masks) equals the oracle on sampled pixels and
    whole rows, asking for the z / P rasters does not change it, and the rasters themselves do not
    depend on the threshold, bit for bit (the chain form's forward pass below the sparse regime; in it the
    register-retaining pass A or -- 80 / 112 float32 and 40 float64 dates -- the time-split pass A, whose
    waves hand the reference's forward fold from slice to slice: the same fold and the same chi-square
    evaluation)."""
    import torch
    from nd_amd import kernels, synth as dsynth
    from oracle import checks
    dev = torch.device('cuda:0')
    ny, nx = 1536, 2048
    st = dsynth.wishart_c2_stack(k, ny, nx, looks=9, seed=5 + k, device=dev, change_frac=0.02)
    st = st.to(getattr(torch, dtype))
    ch = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.01, n=9)
    ch_s, z, P = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.01, n=9, stats=True)
    _, z2, P2 = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9, stats=True)
    torch.cuda.synchronize()
    assert torch.equal(ch, ch_s)
    assert torch.equal(z, z2) and torch.equal(P, P2)
    res = checks.omnibus_sample(st, ch, 0.01, 9, nsample=3000, rows=(0, ny - 1), seed=3)
    assert res['bad'] == 0, res
    assert res['flagged_fraction'] > 0.5


_RASTER_SCRIPT = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
for k, dt in ((24, torch.float32), (16, torch.float32), (12, torch.float64),
              # (round 6) long series: the rasters from the chain form -- in registers for 17 .. 24 float64 dates,
              # in two streaming passes beyond -- at every low threshold
              (24, torch.float64), (40, torch.float32), (96, torch.float32), (40, torch.float64)):
    ny = 1536 if k <= 40 else 512
    st = synth.wishart_c2_stack(k, ny, 2048, looks=9, seed=40 + k, device=dev, change_frac=0.02).to(dt)
    for alpha in (1e-4, 0.01, 0.3):
        ch, z, P = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9, stats=True)
        ch0 = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
        assert torch.equal(ch, ch0)
        h = hashlib.sha1()
        for t in (ch, z, P):
            h.update(t.cpu().numpy().tobytes())
        print('RASTERS', k, str(dt), alpha, h.hexdigest())
    del st
# the fused multilooking kernel: rasters from the kernel itself against the separate pass
st = synth.wishart_c2_stack(12, 768, 2048, looks=4, seed=77, device=dev, change_frac=0.02)
for ml in (3, 5):
    for alpha in (0.01, 0.3):
        ch, z, P = kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=alpha, ml=ml, stats=True)
        ch0 = kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=alpha, ml=ml)
        assert torch.equal(ch, ch0)
        h = hashlib.sha1()
        for t in (ch, z, P):
            h.update(t.cpu().numpy().tobytes())
        print('RASTERS ml', ml, alpha, h.hexdigest())
'''


def test_rasters_from_chain_form_equal_separate_pass():
    """Below the sparse regime the z / P rasters come from the chain form -- its retained series (round 4), for
    long series the forward pass of its two streaming passes, and the fused multilooking kernel's retained
    multilooked series (round 6): no separate read of the planes; ND_AMD_STATS_SPLIT=1 restores the separate
    pass A in front of the search.  Both give
    the same map and bit-identical rasters, at the thresholds of the streaming and of the chain form, on a
    raster large enough for the device-side density gate (each mode in a fresh process: the switch is read
    once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for split in ('0', '1'):
        e = dict(os.environ, ND_AMD_STATS_SPLIT=split)
        r = subprocess.run([sys.executable, '-c', _RASTER_SCRIPT % root], env=e, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('RASTERS')])
    assert len(outs[0]) == 25 and outs[0] == outs[1], (outs[0], outs[1])


@pytest.mark.parametrize('ml', [3, 5])
def test_fused_multilook_whole_raster(oracle, stack, ml):
    """OmnibusTest(ml=w) at full size: the fused kernel's map equals the boxcar kernel followed by the
    plain test on the WHOLE raster at the benchmark's, the reference's default and the tutorial's
    thresholds, and the oracle's (scipy boxcar -> change detection, n = ml ** 2) on four row bands
    including both edges of the raster."""
    import torch
    from nd_amd import kernels
    from oracle import checks
    kern = (np.ones((ml, ml)) / ml ** 2).reshape(1, 1, ml, ml)
    mlk = kernels.convolve(stack, kern)
    for alpha in (0.99, 0.01, 1e-4):
        got = kernels.change_detection_multilooked(stack[0], stack[1], stack[2], stack[3], alpha=alpha, ml=ml)
        assert got is not None
        two = kernels.change_detection(mlk[0], mlk[1], mlk[2], mlk[3], alpha=alpha, n=ml * ml)
        assert torch.equal(got, two), 'alpha = %g' % alpha
        res = checks.omnibus_ml_bands(stack, got, ml, alpha, [(0, 13), (1365, 12), (2049, 11), (NY - 13, 13)])
        assert res['bad'] == 0 and res['compared'] == 49 * NX * K
        del got, two
