"""
examples/nd_binding.py -- the ctypes stub a maintainer of jnhansen/nd would add (as `nd/_amd.py`)
to run the four native call sites of the per-pixel path on libnd_amd.so.  numpy in, numpy out:
exactly the arguments the reference passes today.

    nd/change.py:69        _change.change_detection(values, alpha, n, njobs)      -> change_detection
    nd/filters.py:262-267  snf.convolve(arr, nd_kernel, output=output, **kwargs)  -> convolve
    nd/filters.py:462      _pixelwise_nlmeans_3d(values, _out, r, f, sigma, h, n_eff)
    nd/filters.py:372-378  snf.gaussian_filter(arr, sigma=ndsigma, output=output) -> gaussian_filter

INTEGRATION.md quotes the sections between the `# --8<--` marks verbatim
(tests/test_host_logic.py::test_integration_doc_quotes_the_example checks that), and
tests/test_binding_example_gpu.py runs every function against the oracle / scipy on the GPU.
PyTorch only supplies device memory and the stream.
"""
# --8<-- [load]
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = C.CDLL(os.environ.get('ND_AMD_LIB') or os.path.join(_HERE, '..', 'nd_amd', 'libnd_amd.so'))
assert _L.nd_amd_abi_version() == 1               # built with: python -m nd_amd.build
_L.nd_amd_last_error.restype = C.c_char_p
_L.nd_amd_omnibus_c2_workspace_bytes.restype = C.c_size_t
_L.nd_amd_omnibus_c2_ml_workspace_bytes.restype = C.c_size_t
_DT = {np.dtype('float32'): 0, np.dtype('float64'): 1}            # ND_AMD_F32 / ND_AMD_F64
_MODES = {'reflect': 0, 'constant': 1, 'nearest': 2, 'mirror': 3, 'wrap': 4}


def _check(rc):
    if rc == -4:
        raise ValueError('No solution')                            # find_weight, nd/_filters.pyx:311
    if rc != 0:
        raise RuntimeError(_L.nd_amd_last_error().decode())


def _p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def _i64(v):
    return (C.c_int64 * len(v))(*map(int, v))


def _f64(v):
    return (C.c_double * len(v))(*map(float, v))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
# --8<-- [end]


# --8<-- [omnibus]
def change_detection(c11, c12, c22, alpha, n=1):
    """c11, c22: float32 (y, x, time); c12: complex64 (y, x, time) -- the variables of the dataset
    as nd/change.py:59-66 sees them; returns uint8 (y, x, time) like nd._change.change_detection."""
    ny, nx, k = c11.shape
    d11, d22 = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (c11, c22))
    d12 = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(c12)).cuda())   # (y, x, time, 2)
    change = torch.empty((ny, nx, k), dtype=torch.uint8, device='cuda')
    nbytes = _L.nd_amd_omnibus_c2_workspace_bytes(0, C.c_int64(ny), C.c_int64(nx), C.c_int64(k), None)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    _check(_L.nd_amd_omnibus_c2_pixel_major(
        _p(d11), C.c_void_p(d12.data_ptr()), C.c_void_p(d12.data_ptr() + 4), _p(d22), 0,
        C.c_int64(ny), C.c_int64(nx), C.c_int64(k), _i64((1, 2, 2, 1)),      # date strides per variable
        C.c_uint32(n), C.c_double(alpha), _p(change), None, None,
        _p(ws), C.c_size_t(nbytes), _stream()))
    return change.cpu().numpy()
# --8<-- [end]


# --8<-- [omnibus_ml]
def change_detection_multilooked(c11, c12re, c12im, c22, alpha, ml):
    """nd/change.py:61-69 with ml given: BoxcarFilter(w=ml) over every (y, x) plane, n = ml ** 2, then
    the test -- in one pass.  c11 ... c22: float32 (time, y, x) arrays (the dataset's variables,
    complex C12 already split by disassemble_complex); returns uint8 (y, x, time), or None where the
    fused kernel does not apply (then: BoxcarFilter as before, and change_detection above)."""
    k, ny, nx = c11.shape
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (c11, c12re, c12im, c22)]
    nbytes = _L.nd_amd_omnibus_c2_ml_workspace_bytes(0, C.c_int64(ny), C.c_int64(nx), C.c_int64(k), C.c_int(ml))
    if nbytes == 0:
        return None
    change = torch.empty((ny, nx, k), dtype=torch.uint8, device='cuda')
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    _check(_L.nd_amd_omnibus_c2_ml(
        _p(dev[0]), _p(dev[1]), _p(dev[2]), _p(dev[3]), 0, C.c_int64(ny), C.c_int64(nx), C.c_int64(k),
        C.c_int64(nx), C.c_int64(1), C.c_int64(ny * nx),                       # element strides y, x, time
        C.c_int(ml), C.c_double(alpha), _p(change), None, None, _p(ws), C.c_size_t(nbytes), _stream()))
    return change.cpu().numpy()
# --8<-- [end]


# --8<-- [convolve]
def convolve(arr, nd_kernel, output, mode='reflect', cval=0.0, origin=0):
    """scipy.ndimage.convolve(arr, nd_kernel, output=output, mode=..., cval=..., origin=...) for a
    real array of up to four dimensions (complex arrays keep the reference's own split into
    np.real / np.imag, nd/filters.py:261-265)."""
    from nd_amd.kernels import footprint                           # scipy's tap order, host side
    offs, w = footprint(nd_kernel, origin)                         # (ntaps, ndim) int64, (ntaps,) f64
    pad = 4 - arr.ndim
    a = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    o = torch.empty_like(a)
    offs4 = np.zeros((len(w), 4), np.int64)
    offs4[:, pad:] = offs
    taps = torch.empty(max(24 * len(w), 1), dtype=torch.uint8, device='cuda')   # used beyond 128 taps
    _check(_L.nd_amd_correlate(
        _p(a), _p(o), _DT[arr.dtype], _i64((1,) * pad + tuple(arr.shape)),
        _i64((0,) * pad + tuple(a.stride())), _i64((0,) * pad + tuple(o.stride())), C.c_int64(len(w)),
        offs4.ctypes.data_as(C.POINTER(C.c_int64)), w.ctypes.data_as(C.POINTER(C.c_double)),
        _MODES[mode], C.c_double(cval), _p(taps), C.c_size_t(taps.numel()), _stream()))
    output[...] = o.cpu().numpy()
# --8<-- [end]


# --8<-- [nlmeans]
def _pixelwise_nlmeans_3d(values, out, r, f, sigma, h, n_eff=-1):
    """nd._filters._pixelwise_nlmeans_3d(values, out, r, f, sigma, h, n_eff): values / out are
    (N0, N1, N2, variables) arrays, r / f three unsigned radii."""
    a = torch.from_numpy(np.ascontiguousarray(values)).cuda()
    o = torch.empty_like(a)
    status = torch.zeros(1, dtype=torch.int32, device='cuda')
    N = a.shape[:3]
    _check(_L.nd_amd_nlmeans3d(
        _p(a), _p(o), _DT[values.dtype], _i64(N), C.c_int64(a.shape[3]),
        _i64(a.stride()), _i64(o.stride()),
        (C.c_uint32 * 3)(*map(int, r)), (C.c_uint32 * 3)(*map(int, f)),
        C.c_double(sigma), C.c_double(h), C.c_double(n_eff),
        0,                      # patch_mode 0: what the compiled reference computes
        1, _p(status),          # neff_policy 1: report 'No solution' (what a Cython >= 3 build raises)
        None, None, None, None,             # global_N, tile_off, core_lo, core_hi: NULL = the plain call
        _stream()))
    if int(status.item()):
        raise ValueError('No solution')
    out[...] = o.cpu().numpy()
# --8<-- [end]


# --8<-- [gaussian]
def gaussian_filter(arr, sigma, output, mode='reflect', cval=0.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter(arr, sigma=sigma, output=output, mode=..., truncate=...) for a
    real array of up to four dimensions: one correlate1d pass per axis with sigma > 1e-15, each
    reading the previous one's result in the array dtype (scipy/ndimage/_filters.py gaussian_filter)."""
    sig = [float(sigma)] * arr.ndim if np.isscalar(sigma) else [float(s) for s in sigma]
    pad = 4 - arr.ndim
    src = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    bufs = [torch.empty_like(src), torch.empty_like(src)]
    dims = _i64((1,) * pad + tuple(arr.shape))
    strides = _i64((0,) * pad + tuple(src.stride()))                # all three tensors are contiguous
    done = 0
    for axis, s in enumerate(sig):
        if s <= 1e-15:
            continue
        radius = int(truncate * s + 0.5)
        x = np.arange(-radius, radius + 1)
        w = np.exp(-0.5 / (s * s) * x ** 2)
        w = (w / w.sum())[::-1]                                     # _gaussian_kernel1d(s, 0, radius)[::-1]
        dst = bufs[done % 2]
        _check(_L.nd_amd_correlate1d(
            _p(src), _p(dst), _DT[arr.dtype], dims, strides, strides,
            pad + axis, len(w), _f64(w), _MODES[mode], C.c_double(cval), _stream()))
        src = dst
        done += 1
    output[...] = src.cpu().numpy()
# --8<-- [end]
