#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- generate the golden input/output vectors.

Run in the build container only (needs /root/reference for the nlmeans vectors):

    python oracle/build_ref.py && python tests/golden/make_golden.py

What each file holds and where the expected outputs come from:

  nlmeans_ref.npz   inputs + outputs of the REAL reference kernel
                    nd._filters._pixelwise_nlmeans_3d (nd/_filters.pyx:320-420), compiled
                    unmodified into oracle/_ref by oracle/build_ref.py.  Cases follow the
                    reference's own tests (nd/tests/test_nlmeans_filter.py) and SURVEY 8c F5.
                    `pm1_*` pins the signed-patch semantics (patch_mode 1) through the
                    reference itself: with f = 0 and one channel per patch offset the
                    reference's variable loop IS the patch loop (interior pixels only).
  convolve_scipy.npz inputs + outputs of scipy.ndimage.convolve (scipy is the reference's
                    arithmetic for ConvolutionFilter/BoxcarFilter, nd/filters.py:256-267;
                    cases from nd/tests/test_convolution_filter.py and SURVEY 8c F6).
  omnibus_kat.npz   the reference's known-answer test inputs (nd/tests/test_change_omnibus.py:6-19,
                    rebuilt from the seeds of nd/testing.py:34-70) with the properties that
                    test asserts, plus per-pixel z / P anchors.  nd/_change.pyx cannot be built
                    here (GSL absent), so the anchors come from the oracle and are cross-checked
                    against the values recorded in SURVEY.md 8c (from the survey's own probe)
                    and against scipy.stats.chi2.cdf.

Only data is stored: no reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))

from tests import synth  # noqa: E402


def ref_nlmeans(arr, r, f, sigma, h, n_eff=-1):
    from nd import _filters as RF   # oracle/_ref/nd/_filters*.so == the reference
    out = np.empty_like(arr)
    RF._pixelwise_nlmeans_3d(arr, out, np.array(r, np.uint32), np.array(f, np.uint32),
                             sigma, h, n_eff)
    return out


def make_nlmeans():
    d = {}
    cases = []
    # nd/tests/test_nlmeans_filter.py: ds = generate_test_dataset(y=20, x=20, time=10), 4 variables
    ds = synth.reference_test_dataset({'y': 20, 'x': 20, 'time': 10}, 0, 1)
    arr = np.stack([ds[v] for v in ('C11', 'C12__im', 'C12__re', 'C22')], axis=-1)   # (y,x,time,var)
    arr = np.ascontiguousarray(arr[:12, :14, :4])      # keep the file small
    cases.append(('ds_r110_f110', arr, (1, 1, 0), (1, 1, 0), 2.0, 2.0, -1))       # test_reduce_std
    cases.append(('ds_r000', arr, (0, 0, 0), (0, 0, 0), 1.0, 1.0, -1))            # zero radius
    cases.append(('ds_r110_f000', arr, (1, 1, 0), (0, 0, 0), 0.5, 0.7, -1))       # pixelwise weights
    cases.append(('ds_r221_f000_neff', arr, (2, 2, 1), (0, 0, 0), 1.0, 2.0, 3.0))
    rng = np.random.default_rng(5)
    a32 = rng.gamma(4.0, 0.25, (40, 40, 2, 1)).astype(np.float32)
    cases.append(('f32_r10_f3', a32, (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1))       # config C-C shape
    cases.append(('f32_r3_f0', a32, (3, 3, 0), (0, 0, 0), 0.3, 0.25, -1))
    a4 = rng.gamma(4.0, 0.25, (5, 16, 18, 4)).astype(np.float32)                  # (time,y,x,var)
    cases.append(('f32_tyx_r133_f111', a4, (1, 3, 3), (1, 1, 1), 0.5, 0.5, -1))   # tutorial params
    cases.append(('f32_tyx_r133_f000_neff5', a4, (1, 3, 3), (0, 0, 0), 0.5, 2.0, 5.0))
    for name, a, r, f, s, h, ne in cases:
        d[name + '__in'] = a
        d[name + '__par'] = np.array(list(r) + list(f) + [s, h, ne], np.float64)
        d[name + '__out'] = ref_nlmeans(a, r, f, s, h, ne)
    # patch_mode 1 pin: shifted-channel construction through the real reference
    a = rng.normal(1.0, 0.5, (18, 20)).astype(np.float32)
    fy, fx, ry, rx = 1, 2, 2, 3
    N0, N1 = a.shape

    def R(i, N):
        return -i if i < 0 else (2 * N - 2 - i if i >= N else i)
    chans = []
    for dy in range(-fy, fy + 1):
        for dx in range(-fx, fx + 1):
            ch = np.empty_like(a)
            for y in range(N0):
                for x in range(N1):
                    ch[y, x] = a[R(y + dy, N0), R(x + dx, N1)]
            chans.append(ch)
    stack = np.ascontiguousarray(np.stack(chans, axis=-1)[:, :, None, :])
    ref = ref_nlmeans(stack, (ry, rx, 0), (0, 0, 0), 0.3, 0.5)
    centre = (2 * fy + 1) * (2 * fx + 1) // 2
    d['pm1_in'] = a
    d['pm1_par'] = np.array([ry, rx, 0, fy, fx, 0, 0.3, 0.5, -1], np.float64)
    d['pm1_out_interior'] = ref[ry + fy:-(ry + fy), rx + fx:-(rx + fx), 0, centre]
    np.savez_compressed(os.path.join(HERE, 'nlmeans_ref.npz'), **d)
    print('nlmeans_ref.npz:', len(cases) + 1, 'cases')


def make_convolve():
    import scipy.ndimage as ndi
    d = {}
    ds = synth.reference_test_dataset({'y': 20, 'x': 20, 'time': 10}, 0, 1)
    c11 = ds['C11'][:, :, :3]
    np.random.seed(42)
    k55 = np.random.rand(5, 5)                               # test_convolve_dataset
    ident = np.zeros((3, 3)); ident[1, 1] = 1
    cases = [
        ('rand5x5_f64', c11, k55.reshape(5, 5, 1), {}),
        ('ident_f64', c11, ident.reshape(3, 3, 1), {}),
        ('box3_f64', c11, (np.ones((3, 3)) / 9).reshape(3, 3, 1), {}),
        ('box5_f32', c11.astype(np.float32), (np.ones((5, 5)) / 25).reshape(5, 5, 1), {}),
        ('even4x2_f32', c11.astype(np.float32), np.arange(1, 9, dtype=float).reshape(4, 2, 1) / 36, {}),
        ('box3d_f32', c11.astype(np.float32), np.ones((3, 3, 3)) / 27, {}),
        ('rand5x5_constant', c11, k55.reshape(5, 5, 1), {'mode': 'constant', 'cval': 0.5}),
        ('rand5x5_nearest', c11, k55.reshape(5, 5, 1), {'mode': 'nearest'}),
        ('rand5x5_mirror', c11, k55.reshape(5, 5, 1), {'mode': 'mirror'}),
        ('rand5x5_wrap', c11, k55.reshape(5, 5, 1), {'mode': 'wrap'}),
        ('big9x9_small_arr', c11[:4, :5, :1], np.random.rand(9, 9, 1), {}),     # kernel wider than the array
        ('xy_order_f64', np.ascontiguousarray(c11.transpose(2, 1, 0)), k55.T.reshape(1, 5, 5), {}),
    ]
    for name, a, k, kw in cases:
        d[name + '__in'] = a
        d[name + '__k'] = k
        d[name + '__mode'] = np.array(kw.get('mode', 'reflect'))
        d[name + '__cval'] = np.array(kw.get('cval', 0.0))
        d[name + '__out'] = ndi.convolve(a, k, **kw)
    # complex path of nd/filters.py:261-265: real and imaginary parts separately
    z = (ds['C12__re'] + 1j * ds['C12__im'])[:, :, :2].astype(np.complex64)
    out = np.empty_like(z)
    ndi.convolve(np.real(z), k55.reshape(5, 5, 1), output=np.real(out))
    ndi.convolve(np.imag(z), k55.reshape(5, 5, 1), output=np.imag(out))
    d['complex64__in'] = z
    d['complex64__k'] = k55.reshape(5, 5, 1)
    d['complex64__out'] = out
    np.savez_compressed(os.path.join(HERE, 'convolve_scipy.npz'), **d)
    print('convolve_scipy.npz:', len(cases) + 1, 'cases')


def make_omnibus():
    from oracle import oracle as O
    d = {}
    dims = {'y': 5, 'x': 5, 'time': 10}
    d1 = synth.reference_test_dataset(dims, [1, 0, 0, 1], 0.1)
    d2 = synth.reference_test_dataset(dims, [10, 0, 0, 10], 0.1)
    ds = {v: np.concatenate([d1[v][..., :5], d2[v][..., 5:]], axis=2) for v in d1}
    values = np.stack([ds['C11'], ds['C12__re'], ds['C12__im'], ds['C22']], axis=-1)   # (y,x,t,4)
    d['kat_values'] = values
    for dt, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
        ch, z, P = O.change_detection(values.astype(dt), 0.9, 9, stats=True)
        assert ch[:, :, 5].all() and (ch.sum(axis=2) == 1).all()
        d['kat_change_' + tag] = ch
        d['kat_z_' + tag] = z
        d['kat_P_' + tag] = P
        Pf5, zf5 = O.single_pixel_omnibus(values[0, 0, :5].astype(dt), 9)
        d['kat_first5_' + tag] = np.array([Pf5, zf5], np.float64)
    # NaN path of nd/tests/test_change_common.py:21-32 (default OmnibusTest(): n=1, alpha=0.01)
    dn = synth.reference_test_dataset({'y': 20, 'x': 30, 'time': 10}, 0, 1)
    vn = np.stack([dn['C11'], dn['C12__re'], dn['C12__im'], dn['C22']], axis=-1)[:8, :8]
    chn, zn, Pn = O.change_detection(vn, 0.01, 1, stats=True)
    d['nan_values'] = vn
    d['nan_change'] = chn
    d['nan_z'] = zn
    d['nan_P'] = Pn
    # Wishart stacks with step changes, small
    planes = synth.omnibus_stack(seed=21, k=12, ny=16, nx=24, dtype=np.float32, change_frac=0.15)
    vw = np.ascontiguousarray(np.stack([np.moveaxis(p, 0, -1) for p in planes], axis=-1))
    d['wishart_values_f32'] = vw
    for alpha in (0.9, 0.99, 0.9999):
        ch, z, P = O.change_detection(vw, alpha, 9, stats=True)
        tag = ('%g' % alpha).replace('.', 'p')
        d['wishart_change_' + tag] = ch
        if alpha == 0.9:
            d['wishart_z'] = z
            d['wishart_P'] = P
    np.savez_compressed(os.path.join(HERE, 'omnibus_kat.npz'), **d)
    print('omnibus_kat.npz written')


if __name__ == '__main__':
    make_nlmeans()
    make_convolve()
    make_omnibus()
