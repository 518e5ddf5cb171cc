"""
oracle/oracle.py -- ctypes front end of the CPU oracle (liboracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by anything under nd_amd/.

The numpy-facing signatures mirror the reference's native entry points:

  change_detection(values, alpha, n, njobs)   <- nd/_change.pyx:263-287
  single_pixel_omnibus(ts, n)                 <- nd/_change.pyx:133-151
  pixelwise_nlmeans_3d(arr, output, r, f, sigma, h, n_eff)
                                              <- nd/_filters.pyx:320-420
  convolve(input, weights, output, mode, cval, origin)
                                              <- scipy.ndimage.convolve as called
                                                 at nd/filters.py:256-267
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64 = C.c_int64
_dbl = C.c_double
_vp = C.c_void_p
_MODES = {'reflect': 0, 'constant': 1, 'nearest': 2, 'mirror': 3, 'wrap': 4,
          'grid-constant': 1, 'grid-wrap': 4, 'grid-mirror': 0}


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('nd_oracle.c', 'nd_oracle_impl.h')]
    if (not force and os.path.exists(so)
            and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs)):
        return so
    subprocess.check_call(['make', '-C', _HERE, 'liboracle.so', '-B'],
                          stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        for name in ('oracle_cdf_chisq_P', 'oracle_gammainc_P', 'oracle_gammainc_Q'):
            getattr(L, name).restype = _dbl
            getattr(L, name).argtypes = [_dbl, _dbl]
        for name in ('oracle_f', 'oracle_rho'):
            getattr(L, name).restype = _dbl
            getattr(L, name).argtypes = [_dbl, _dbl, _dbl]
        L.oracle_omega2.restype = _dbl
        L.oracle_omega2.argtypes = [_dbl] * 4
        L.oracle_find_weight.restype = _dbl
        L.oracle_find_weight.argtypes = [_dbl, _dbl, _dbl, C.POINTER(C.c_int)]
        L.oracle_extend.restype = _i64
        L.oracle_extend.argtypes = [_i64, _i64, C.c_int]
        for sfx in ('f32', 'f64'):
            fn = getattr(L, 'oracle_omnibus_c2_' + sfx)
            fn.restype = C.c_int
            fn.argtypes = [_vp] * 4 + [_i64] * 6 + [C.c_uint, _dbl, _vp, _vp, _vp, C.c_int]
            fn = getattr(L, 'oracle_omnibus_pol_' + sfx)
            fn.restype = C.c_int
            fn.argtypes = [_vp, C.c_int] + [_i64] * 6 + [C.c_uint, _dbl, _vp, _vp, _vp, C.c_int]
            fn = getattr(L, 'oracle_single_pixel_omnibus_' + sfx)
            fn.restype = C.c_float if sfx == 'f32' else _dbl
            fn.argtypes = [_vp, _i64, C.c_uint, _vp]
            fn = getattr(L, 'oracle_nlmeans3d_' + sfx)
            fn.restype = C.c_int
            fn.argtypes = [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl,
                           C.c_int, C.c_int, C.c_int]
            fn = getattr(L, 'oracle_correlate_fp_' + sfx)
            fn.restype = C.c_int
            fn.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int]
            fn = getattr(L, 'oracle_correlate_mode_' + sfx)
            fn.restype = C.c_int
            fn.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int, _dbl]
        _LIB = L
    return _LIB


def _sfx(dtype):
    if dtype == np.float32:
        return 'f32'
    if dtype == np.float64:
        return 'f64'
    raise TypeError('oracle supports float32/float64, got %r' % (dtype,))


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _estr(a):
    """element strides of a numpy array"""
    assert all(s % a.itemsize == 0 for s in a.strides)
    return [s // a.itemsize for s in a.strides]


# --------------------------------------------------------------------------
# scalars
# --------------------------------------------------------------------------
def cdf_chisq_P(x, nu):
    return lib().oracle_cdf_chisq_P(float(x), float(nu))


def gammainc_P(a, x):
    return lib().oracle_gammainc_P(float(a), float(x))


def gammainc_Q(a, x):
    return lib().oracle_gammainc_Q(float(a), float(x))


def f_dof(p, k, n):
    return lib().oracle_f(float(p), float(k), float(n))


def rho(p, k, n):
    return lib().oracle_rho(float(p), float(k), float(n))


def omega2(p, k, n, rho_):
    return lib().oracle_omega2(float(p), float(k), float(n), float(rho_))


def find_weight(W, W2, n):
    err = C.c_int(0)
    w = lib().oracle_find_weight(float(W), float(W2), float(n), C.byref(err))
    if err.value == 1:
        raise ValueError('No solution')
    if err.value == 2:
        raise ZeroDivisionError('float division')
    return w


# --------------------------------------------------------------------------
# omnibus
# --------------------------------------------------------------------------
def single_pixel_omnibus(ts, n):
    """(P, z) of one k x 4 series [C11, C12re, C12im, C22]; nd/_change.pyx:133-151."""
    ts = np.ascontiguousarray(ts)
    sfx = _sfx(ts.dtype)
    assert ts.ndim == 2 and ts.shape[1] == 4
    z = np.zeros(1, ts.dtype)
    P = getattr(lib(), 'oracle_single_pixel_omnibus_' + sfx)(
        _ptr(ts), ts.shape[0], int(n), _ptr(z))
    return ts.dtype.type(P), z[0]


def change_detection(values, alpha, n=1, njobs=1, stats=False):
    """values: (y, x, time, 4) float32/float64 array (any strides), columns
    [C11, C12re, C12im, C22].  Returns uint8 (y, x, time); with stats=True
    also the global-test z and P rasters (y, x)."""
    values = np.asarray(values)
    sfx = _sfx(values.dtype)
    assert values.ndim == 4 and values.shape[3] == 4
    ny, nx, k, _ = values.shape
    planes = [values[..., v] for v in range(4)]
    return change_detection_planes(planes, alpha, n, njobs, stats)


def change_detection_planes(planes, alpha, n=1, njobs=1, stats=False):
    """planes: 4 arrays (y, x, time) sharing dtype and strides."""
    p0 = planes[0]
    sfx = _sfx(p0.dtype)
    ny, nx, k = p0.shape
    es = _estr(p0)
    for p in planes:
        assert p.shape == p0.shape and p.dtype == p0.dtype and _estr(p) == es
    change = np.zeros((ny, nx, k), np.uint8)
    z = np.zeros((ny, nx), p0.dtype)
    P = np.zeros((ny, nx), p0.dtype)
    rc = getattr(lib(), 'oracle_omnibus_c2_' + sfx)(
        _ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2]), _ptr(planes[3]),
        ny, nx, k, es[0], es[1], es[2], int(n), float(alpha),
        _ptr(change), _ptr(z), _ptr(P), int(njobs))
    if rc != 0:
        raise RuntimeError('oracle_omnibus_c2 failed: %d' % rc)
    if stats:
        return change, z, P
    return change


def change_detection_pol(planes, pol, alpha, n=1, njobs=1, stats=False):
    """Generic-p omnibus (p = 2 or 3) on p*p planes (y, x, time) sharing dtype and strides.
    p = 3 plane order: C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im.  There is no
    reference implementation for p = 3 (parity unpinned); p = 2 must equal change_detection."""
    assert len(planes) == pol * pol
    p0 = planes[0]
    sfx = _sfx(p0.dtype)
    ny, nx, k = p0.shape
    es = _estr(p0)
    for p in planes:
        assert p.shape == p0.shape and p.dtype == p0.dtype and _estr(p) == es
    ptrs = (C.c_void_p * len(planes))(*[p.ctypes.data for p in planes])
    change = np.zeros((ny, nx, k), np.uint8)
    z = np.zeros((ny, nx), p0.dtype)
    P = np.zeros((ny, nx), p0.dtype)
    rc = getattr(lib(), 'oracle_omnibus_pol_' + sfx)(
        C.cast(ptrs, C.c_void_p), int(pol), ny, nx, k, es[0], es[1], es[2], int(n), float(alpha),
        _ptr(change), _ptr(z), _ptr(P), int(njobs))
    if rc != 0:
        raise RuntimeError('oracle_omnibus_pol failed: %d' % rc)
    if stats:
        return change, z, P
    return change


# --------------------------------------------------------------------------
# nlmeans
# --------------------------------------------------------------------------
def pixelwise_nlmeans_3d(arr, output, r, f, sigma, h, n_eff=-1, neff_policy=1,
                         njobs=1, patch_mode=0):
    """In-place into `output`, like nd/_filters.pyx:320-325.

    patch_mode 0 = the compiled reference on LP64 (patch loops are empty when
    any f > 0, see nd_oracle_impl.h); 1 = signed patch range as the source
    text intends."""
    assert arr.ndim == 4 and output.shape == arr.shape and output.dtype == arr.dtype
    sfx = _sfx(arr.dtype)
    N = np.array(arr.shape[:3], np.int64)
    s = np.array(_estr(arr), np.int64)
    so = np.array(_estr(output), np.int64)
    r = np.ascontiguousarray(r, np.uint32)
    f = np.ascontiguousarray(f, np.uint32)
    assert r.shape == (3,) and f.shape == (3,)
    rc = getattr(lib(), 'oracle_nlmeans3d_' + sfx)(
        _ptr(arr), _ptr(output), _ptr(N), arr.shape[3], _ptr(s), _ptr(so),
        _ptr(r), _ptr(f), float(sigma), float(h), float(n_eff), int(neff_policy),
        int(patch_mode), int(njobs))
    if rc == 1:
        raise ValueError('No solution')
    if rc != 0:
        raise RuntimeError('oracle_nlmeans3d failed: %d' % rc)


# --------------------------------------------------------------------------
# convolution (scipy.ndimage.convolve semantics)
# --------------------------------------------------------------------------
def footprint(weights, origin=0, convolution=True):
    """Non-zero taps of the kernel as scipy's NI_Correlate sees them:
    returns (offsets[ntaps, ndim] int64, w[ntaps] float64) in footprint
    (row-major) order.  scipy/ndimage/_filters.py `_correlate_or_convolve`."""
    weights = np.asarray(weights, dtype=np.float64)
    nd_ = weights.ndim
    origins = [origin] * nd_ if np.isscalar(origin) else list(origin)
    if convolution:
        weights = weights[tuple([slice(None, None, -1)] * nd_)]
        for ii in range(nd_):
            origins[ii] = -origins[ii]
            if not weights.shape[ii] & 1:
                origins[ii] -= 1
    for o, lenw in zip(origins, weights.shape):
        if (lenw // 2 + o < 0) or (lenw // 2 + o >= lenw):
            raise ValueError('invalid origin')
    offs, w = [], []
    for idx in np.ndindex(*weights.shape):
        v = weights[idx]
        if abs(v) > np.finfo(np.float64).eps:
            offs.append([idx[d] - (weights.shape[d] // 2 + origins[d]) for d in range(nd_)])
            w.append(v)
    return (np.array(offs, np.int64).reshape(len(w), nd_),
            np.array(w, np.float64))


def convolve(input, weights, output=None, mode='reflect', cval=0.0, origin=0):
    """scipy.ndimage.convolve restated (real dtypes, ndim <= 4)."""
    input = np.asarray(input)
    if np.iscomplexobj(input):
        raise TypeError('complex input: call on .real and .imag like nd/filters.py:261-265')
    sfx = _sfx(input.dtype)
    weights = np.asarray(weights, np.float64)
    if weights.ndim != input.ndim:
        raise RuntimeError('filter weights array has incorrect shape.')
    if input.ndim > 4:
        raise NotImplementedError
    if output is None:
        output = np.empty_like(input)
    offs, w = footprint(weights, origin, convolution=True)
    pad = 4 - input.ndim
    A = np.array((1,) * pad + input.shape, np.int64)
    si = np.array([0] * pad + _estr(input), np.int64)
    so = np.array([0] * pad + _estr(output), np.int64)
    offs4 = np.zeros((len(w), 4), np.int64)
    offs4[:, pad:] = offs
    m = _MODES[mode]
    fn = getattr(lib(), 'oracle_correlate_mode_' + sfx)
    rc = fn(_ptr(input), _ptr(output), _ptr(A), _ptr(si), _ptr(so), len(w),
            _ptr(offs4), _ptr(w), m, float(cval))
    if rc != 0:
        raise RuntimeError('oracle_correlate failed: %d' % rc)
    return output


def convolve_reflect_mt(input, weights, output=None, njobs=1):
    """Multithreaded reflect-mode variant used as the bench CPU baseline."""
    input = np.asarray(input)
    sfx = _sfx(input.dtype)
    if output is None:
        output = np.empty_like(input)
    offs, w = footprint(np.asarray(weights, np.float64), 0, True)
    pad = 4 - input.ndim
    A = np.array((1,) * pad + input.shape, np.int64)
    si = np.array([0] * pad + _estr(input), np.int64)
    so = np.array([0] * pad + _estr(output), np.int64)
    offs4 = np.zeros((len(w), 4), np.int64)
    offs4[:, pad:] = offs
    fn = getattr(lib(), 'oracle_correlate_fp_' + sfx)
    rc = fn(_ptr(input), _ptr(output), _ptr(A), _ptr(si), _ptr(so), len(w),
            _ptr(offs4), _ptr(w), int(njobs))
    if rc != 0:
        raise RuntimeError('oracle_correlate failed: %d' % rc)
    return output


if __name__ == '__main__':
    build(force=True)
    print('built', os.path.join(_HERE, 'liboracle.so'))
    sys.exit(0)
