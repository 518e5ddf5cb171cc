"""
nd_amd/kernels.py -- tensor-level entry points: PyTorch-ROCm tensors in, HIP
kernels launched through the ctypes C ABI (include/nd_amd.h) on torch's current
stream.  PyTorch is only the device-memory / stream plumbing here.

Each function names the native call of the reference it stands in for:

  change_detection(...)      nd._change.change_detection          nd/_change.pyx:263-287
  correlate_footprint(...)   scipy.ndimage.convolve               nd/filters.py:256-267
  pixelwise_nlmeans_3d(...)  nd._filters._pixelwise_nlmeans_3d    nd/_filters.pyx:320-420
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %r' % (name, type(t)))
    if not t.is_cuda:
        raise ValueError('%s must live on a ROCm device (got %s); nd_amd has no CPU path'
                         % (name, t.device))
    if t.dtype not in _DT:
        raise TypeError('%s must be float32 or float64, got %s' % (name, t.dtype))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------------------
# OmnibusTest C2
# ---------------------------------------------------------------------------
def change_detection(c11, c12re, c12im, c22, alpha, n=1, dims=('time', 'y', 'x'),
                     stats=False, workspace='recommended'):
    """Omnibus change detection on four covariance planes.

    c11, c12re, c12im, c22 : CUDA tensors of identical shape/strides/dtype whose
        three axes are named by `dims` (any order of 'y', 'x', 'time'); the
        reference's (y, x, time, 4) view (nd/change.py:66-67) is these four
        tensors side by side.
    workspace : 'recommended' (room for the series of 1/8 of the pixels, so the change-point
        search never re-reads the planes) or 'minimal' (candidate list only: less memory, listed
        pixels are gathered from the planes).
    Returns uint8 tensor (y, x, time) [, z (y, x), P (y, x)].
    """
    planes = (c11, c12re, c12im, c22)
    for name, t in zip(('c11', 'c12re', 'c12im', 'c22'), planes):
        _require_cuda(t, name)
        if t.dim() != 3:
            raise ValueError('%s must be 3-D, got shape %s' % (name, tuple(t.shape)))
        # strides of length-1 axes address nothing: views of one buffer may carry different ones
        same_strides = all(a == b for a, b, n_ in zip(t.stride(), c11.stride(), t.shape) if n_ > 1)
        if (t.shape != c11.shape or not same_strides or t.dtype != c11.dtype
                or t.device != c11.device):
            raise ValueError('the four covariance planes must share shape, strides, '
                             'dtype and device')
    dims = tuple(dims)
    if sorted(dims) != ['time', 'x', 'y']:
        raise ValueError("dims must be a permutation of ('time', 'y', 'x')")
    ay, ax, at = dims.index('y'), dims.index('x'), dims.index('time')
    ny, nx, k = c11.shape[ay], c11.shape[ax], c11.shape[at]
    sy, sx, st = c11.stride(ay), c11.stride(ax), c11.stride(at)
    dev = c11.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
        z = torch.empty((ny, nx), dtype=c11.dtype, device=dev) if stats else None
        P = torch.empty((ny, nx), dtype=c11.dtype, device=dev) if stats else None
        if ny * nx * k > 0:
            min_bytes = C.c_size_t(0)
            nbytes = L.nd_amd_omnibus_c2_workspace_bytes(_DT[c11.dtype], ny, nx, k,
                                                         C.byref(min_bytes))
            if workspace == 'minimal':
                nbytes = min_bytes.value
            elif workspace != 'recommended':
                raise ValueError("workspace must be 'recommended' or 'minimal'")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.nd_amd_omnibus_c2(
                _ptr(c11), _ptr(c12re), _ptr(c12im), _ptr(c22), _DT[c11.dtype],
                ny, nx, k, sy, sx, st, int(n), float(alpha),
                _ptr(change), _ptr(z), _ptr(P), _ptr(ws), nbytes, _stream_ptr(dev)))
            # keep the workspace alive until the stream has consumed it
            ws.record_stream(torch.cuda.current_stream(dev))
    if stats:
        return change, z, P
    return change


def change_detection_multilooked(c11, c12re, c12im, c22, alpha, ml, stats=False):
    """OmnibusTest(ml=w) in one pass over four planar (time, y, x) device planes: boxcar multilooking
    (scipy's arithmetic, 'reflect' border) fused into the test, n = ml ** 2 looks (nd/change.py:61-69).
    Returns None where the fused kernel does not apply (the caller multilooks the stack and uses
    change_detection); otherwise the uint8 map (y, x, time) [, z, P]."""
    planes = (c11, c12re, c12im, c22)
    if not all(torch.is_tensor(t) and t.is_cuda and t.dim() == 3 for t in planes):
        return None
    if c11.dtype != torch.float32 or int(ml) != ml:
        return None
    for t in planes:
        if t.shape != c11.shape or t.stride() != c11.stride() or t.dtype != c11.dtype or t.device != c11.device:
            return None
    k, ny, nx = c11.shape
    st, sy, sx = c11.stride()
    dev = c11.device
    L = _lib.lib()
    nbytes = L.nd_amd_omnibus_c2_ml_workspace_bytes(_lib.F32, ny, nx, k, int(ml)) if ny * nx * k > 0 else 0
    if nbytes == 0 or sx != 1:
        return None
    with torch.cuda.device(dev):
        change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
        z = torch.empty((ny, nx), dtype=c11.dtype, device=dev) if stats else None
        P = torch.empty((ny, nx), dtype=c11.dtype, device=dev) if stats else None
        # The dump of the fused path has a slot for every pixel of the raster (the multilooked series exists
        # nowhere else), about the size of the input, of which only the listed part is ever touched.  torch's
        # caching allocator hands the same block back from call to call; where the memory is not there the
        # caller takes the two-step path (boxcar kernel, then the plain test), which needs none.
        try:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            return None
        rc = L.nd_amd_omnibus_c2_ml(_ptr(c11), _ptr(c12re), _ptr(c12im), _ptr(c22), _lib.F32, ny, nx, k,
                                    sy, sx, st, int(ml), float(alpha), _ptr(change), _ptr(z), _ptr(P),
                                    _ptr(ws), nbytes, _stream_ptr(dev))
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc)
        ws.record_stream(torch.cuda.current_stream(dev))
    return (change, z, P) if stats else change


def change_detection_pixel_major(c11, c12re, c12im, c22, alpha, n=1, stats=False):
    """change_detection for four device variables in the reference's own layout, (y, x, time) with
    time fastest; `c12re` / `c12im` may be the `.real` / `.imag` views of one complex tensor (read
    once).  Returns None when the layout or the series length is not one the pixel-major kernel
    takes (the caller then transposes and uses change_detection)."""
    vs = (c11, c12re, c12im, c22)
    if not all(torch.is_tensor(t) and t.is_cuda and t.dim() == 3 for t in vs):
        return None
    ny, nx, k = c11.shape
    dt = c11.dtype
    if dt not in _DT or k > 192 or ny * nx * k == 0:
        return None
    # beyond the register-retaining sizes the kernel takes the sparse regime only (include/nd_amd.h)
    long_series = k > (24 if dt == torch.float32 else 12)
    ids = []
    for t in vs:
        if t.shape != c11.shape or t.dtype != dt or t.device != c11.device:
            return None
        ids.append(_pixel_major_stride(t))
        if ids[-1] is None:
            return None
    # The date strides come from the tensors' own strides only.  (Two views one element apart are
    # NOT proof of an interleaved complex tensor: buf[:-1] and buf[1:] of a real buffer look the
    # same.  The library pairs the two C12 reads itself when both strides are 2 and the pointers
    # are adjacent.)
    if ids[1] != ids[2]:
        return None
    dev = c11.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
        z = torch.empty((ny, nx), dtype=dt, device=dev) if stats else None
        P = torch.empty((ny, nx), dtype=dt, device=dev) if stats else None
        nbytes = L.nd_amd_omnibus_c2_workspace_bytes(_DT[dt], ny, nx, k, None)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = L.nd_amd_omnibus_c2_pixel_major(
            _ptr(c11), _ptr(c12re), _ptr(c12im), _ptr(c22), _DT[dt], ny, nx, k,
            _lib.i64_array(ids), int(n), float(alpha), _ptr(change), _ptr(z), _ptr(P), _ptr(ws), nbytes,
            _stream_ptr(dev))
        if long_series and rc == _lib.EUNSUPPORTED:
            return None                   # the caller transposes and takes the planar path
        _lib.check(rc)
        ws.record_stream(torch.cuda.current_stream(dev))
    return (change, z, P) if stats else change


def change_detection_c3(planes, alpha, n=1, dims=('time', 'y', 'x'), stats=False):
    """Full-pol (3 x 3) omnibus change detection -- an extension, the reference is dual-pol only.
    planes: nine CUDA tensors [C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im] of
    identical shape/strides/dtype, axes named by `dims`.  Returns uint8 (y, x, time) [, z, P]."""
    planes = list(planes)
    if len(planes) != 9:
        raise ValueError('full-pol covariance needs nine real planes')
    p0 = planes[0]
    for i, t in enumerate(planes):
        _require_cuda(t, 'planes[%d]' % i)
        # strides of length-1 axes address nothing: views of one buffer may carry different ones
        same_strides = t.dim() == 3 and all(a == b for a, b, n_ in zip(t.stride(), p0.stride(), t.shape)
                                            if n_ > 1)
        if (t.dim() != 3 or t.shape != p0.shape or not same_strides
                or t.dtype != p0.dtype or t.device != p0.device):
            raise ValueError('the nine covariance planes must be 3-D and share shape, strides, '
                             'dtype and device')
    dims = tuple(dims)
    if sorted(dims) != ['time', 'x', 'y']:
        raise ValueError("dims must be a permutation of ('time', 'y', 'x')")
    ay, ax, at = dims.index('y'), dims.index('x'), dims.index('time')
    ny, nx, k = p0.shape[ay], p0.shape[ax], p0.shape[at]
    dev = p0.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
        z = torch.empty((ny, nx), dtype=p0.dtype, device=dev) if stats else None
        P = torch.empty((ny, nx), dtype=p0.dtype, device=dev) if stats else None
        if ny * nx * k > 0:
            nbytes = L.nd_amd_omnibus_c3_workspace_bytes(ny, nx, k)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in planes])
            _lib.check(L.nd_amd_omnibus_c3(
                ptrs, _DT[p0.dtype], ny, nx, k, p0.stride(ay), p0.stride(ax), p0.stride(at),
                int(n), float(alpha), _ptr(change), _ptr(z), _ptr(P), _ptr(ws), nbytes,
                _stream_ptr(dev)))
            ws.record_stream(torch.cuda.current_stream(dev))
    if stats:
        return change, z, P
    return change


def change_detection_c3_pixel_major(planes, alpha, n=1, stats=False):
    """change_detection_c3 for nine device variables in the reference's own layout, (y, x, time) with time
    fastest -- [C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im], the off-diagonal pairs either real
    arrays or the `.real` / `.imag` views of complex tensors (read once).  Returns None where the kernel
    does not apply (other layouts, series lengths, thresholds below the sparse regime): the caller then
    transposes and uses change_detection_c3."""
    planes = list(planes)
    if len(planes) != 9 or not all(torch.is_tensor(t) and t.is_cuda and t.dim() == 3 for t in planes):
        return None
    p0 = planes[0]
    ny, nx, k = p0.shape
    dt = p0.dtype
    if dt not in _DT or ny * nx * k == 0:
        return None
    ids = []
    for t in planes:
        if t.shape != p0.shape or t.dtype != dt or t.device != p0.device:
            return None
        ids.append(_pixel_major_stride(t))
        if ids[-1] is None:
            return None
    # the entry point's documented conditions (include/nd_amd.h), checked before anything is allocated: with the
    # reference's default alpha = 0.01 every full-pol (y, x, time) call would otherwise allocate the map and the
    # workspace only to learn that it is declined
    ve = 16 // p0.element_size()
    if (not (alpha >= 0.75) or k % ve != 0 or 16 * 9 * k * p0.element_size() > 56 * 1024
            or any(t.data_ptr() % 16 != 0 for t, i in zip(planes, ids) if i == 1)):
        return None
    dev = p0.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
        z = torch.empty((ny, nx), dtype=dt, device=dev) if stats else None
        P = torch.empty((ny, nx), dtype=dt, device=dev) if stats else None
        nbytes = L.nd_amd_omnibus_c3_workspace_bytes(ny, nx, k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in planes])
        rc = L.nd_amd_omnibus_c3_pixel_major(ptrs, _DT[dt], ny, nx, k, _lib.i64_array(ids), int(n), float(alpha),
                                             _ptr(change), _ptr(z), _ptr(P), _ptr(ws), nbytes, _stream_ptr(dev))
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc)
        ws.record_stream(torch.cuda.current_stream(dev))
    return (change, z, P) if stats else change


# ---------------------------------------------------------------------------
# layout change in front of the hot path
# ---------------------------------------------------------------------------
def _pixel_major_stride(t):
    """1 if the 3-D tensor `t` (y, x, time) is C-contiguous, 2 if it is one half of an interleaved
    complex tensor of that layout, else None.  Strides of length-1 axes carry no information."""
    ny, nx, k = t.shape
    for ids in (1, 2):
        want = (nx * k * ids, k * ids, ids)
        if all(n == 1 or s == w for n, s, w in zip(t.shape, t.stride(), want)):
            return ids
    return None


def _planar_ok(t, ny, nx):
    """(time, y, x) tensor with x contiguous, rows adjacent, any plane pitch >= ny * nx."""
    k = t.shape[0]
    return ((nx == 1 or t.stride(2) == 1) and (ny == 1 or t.stride(1) == nx)
            and (k == 1 or t.stride(0) >= ny * nx))


_RELAYOUT_LDS = 48 * 1024      # bytes of LDS one pixel's series may take in the transpose kernels


def relayout_planar(src, dst):
    """Copy a device variable laid out (y, x, time) with time fastest -- the reference's own
    layout -- into the planar (time, y, x) view `dst` (x fastest, any plane pitch).  `src` may be
    a real tensor, or the `.real` / `.imag` view of an interleaved complex tensor.  Returns False
    (nothing done) when the layouts are not of that form; the caller then copies through torch."""
    if not (torch.is_tensor(src) and src.is_cuda and src.dim() == 3 and dst.dim() == 3):
        return False
    ny, nx, k = src.shape
    if tuple(dst.shape) != (k, ny, nx) or src.dtype != dst.dtype or src.dtype not in _DT:
        return False
    if ny * nx * k == 0:
        return True
    ids = _pixel_major_stride(src)
    if ids is None or not _planar_ok(dst, ny, nx):
        return False
    if (k | 1) * src.element_size() > _RELAYOUT_LDS:        # series too long for the staging buffer
        return False
    dev = src.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nd_amd_relayout_planar(
            _ptr(src), _ptr(dst), _DT[src.dtype], ny * nx, k, ids, max(dst.stride(0), ny * nx),
            _stream_ptr(dev)))
    return True


def relayout_planar_complex(src_re, src_im, dst_re, dst_im):
    """relayout_planar for both halves of one interleaved complex (y, x, time) tensor at once
    (`src_re`, `src_im` = its `.real` / `.imag` views): the source is read once.  False when the two
    views are not the halves of one such tensor."""
    if not (torch.is_tensor(src_re) and torch.is_tensor(src_im) and src_re.is_cuda and src_re.dim() == 3):
        return False
    ny, nx, k = src_re.shape
    es = src_re.element_size()
    if (src_im.shape != src_re.shape or src_re.dtype not in _DT
            or src_im.dtype != src_re.dtype or src_im.data_ptr() != src_re.data_ptr() + es
            or _pixel_major_stride(src_re) is None or _pixel_major_stride(src_im) is None
            or (ny * nx * k > 1 and (_pixel_major_stride(src_re) != 2 or _pixel_major_stride(src_im) != 2))):
        return False
    if ((2 * k) | 1) * es > _RELAYOUT_LDS:
        return False
    for d in (dst_re, dst_im):
        if (tuple(d.shape) != (k, ny, nx) or d.dtype != src_re.dtype or not _planar_ok(d, ny, nx)
                or (k > 1 and d.stride(0) != dst_re.stride(0))):
            return False
    if ny * nx * k == 0:
        return True
    dev = src_re.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nd_amd_relayout_planar_complex(
            _ptr(src_re), _ptr(dst_re), _ptr(dst_im), _DT[src_re.dtype], ny * nx, k,
            max(dst_re.stride(0), ny * nx),
            _stream_ptr(dev)))
    return True


def split_complex(src_re, src_im):
    """The `.real` / `.imag` views of ONE contiguous complex CUDA tensor as two packed real tensors
    of the same shape, in one pass over its memory (nd_amd_split_complex).  None when the views
    are anything else (the caller then packs them one by one)."""
    if not (torch.is_tensor(src_re) and torch.is_tensor(src_im) and src_re.is_cuda
            and src_re.dtype in _DT and src_im.dtype == src_re.dtype and src_re.shape == src_im.shape
            and src_re.device == src_im.device and src_re.numel() > 0):
        return None
    es = src_re.element_size()
    if src_im.data_ptr() != src_re.data_ptr() + es or src_re.stride() != src_im.stride():
        return None
    # a contiguous complex tensor seen through a real view: every stride doubled, innermost 2
    want = []
    acc = 2
    for n_ in reversed(src_re.shape):
        want.append(acc)
        acc *= n_
    want = tuple(reversed(want))
    if any(s_ != w_ for s_, w_, n_ in zip(src_re.stride(), want, src_re.shape) if n_ > 1):
        return None
    if src_re.data_ptr() % 16:
        return None
    dev = src_re.device
    with torch.cuda.device(dev):
        re = torch.empty(src_re.shape, dtype=src_re.dtype, device=dev)
        im = torch.empty(src_re.shape, dtype=src_re.dtype, device=dev)
        _lib.check(_lib.lib().nd_amd_split_complex(_ptr(src_re), _ptr(re), _ptr(im), _DT[src_re.dtype],
                                                   src_re.numel(), _stream_ptr(dev)))
    return re, im


def merge_complex(re, im, out):
    """Two packed real CUDA tensors -> the contiguous complex tensor `out` of the same shape, one
    pass (nd_amd_merge_complex).  False when the tensors are not of that form."""
    if not (torch.is_tensor(out) and out.is_complex() and out.is_cuda and out.is_contiguous()
            and torch.is_tensor(re) and torch.is_tensor(im) and re.is_contiguous() and im.is_contiguous()
            and re.shape == out.shape and im.shape == out.shape and re.dtype in _DT and im.dtype == re.dtype
            and torch.view_as_real(out).dtype == re.dtype and re.device == out.device == im.device
            and out.numel() > 0):
        return False
    if (re.data_ptr() | im.data_ptr() | out.data_ptr()) % 16:
        return False
    dev = out.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nd_amd_merge_complex(_ptr(re), _ptr(im), _ptr(out), _DT[re.dtype],
                                                   re.numel(), _stream_ptr(dev)))
    return True


def relayout_pixel_major(src, dst):
    """Inverse of relayout_planar: planar (time, y, x) `src` (x fastest, any plane pitch) into
    `dst` laid out (y, x, time) with time fastest (a real tensor or one half of a complex one).
    Returns False when the layouts are not of that form."""
    if not (torch.is_tensor(dst) and dst.is_cuda and src.is_cuda and src.dim() == 3 and dst.dim() == 3):
        return False
    ny, nx, k = dst.shape
    if tuple(src.shape) != (k, ny, nx) or src.dtype != dst.dtype or src.dtype not in _DT:
        return False
    if ny * nx * k == 0:
        return True
    ods = _pixel_major_stride(dst)
    if ods is None or not _planar_ok(src, ny, nx):
        return False
    if (k | 1) * src.element_size() > _RELAYOUT_LDS:
        return False
    dev = src.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nd_amd_relayout_pixel_major(
            _ptr(src), _ptr(dst), _DT[src.dtype], ny * nx, k, max(src.stride(0), ny * nx), ods,
            _stream_ptr(dev)))
    return True


# ---------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------
def footprint(weights, origin=0, convolution=True):
    """scipy.ndimage `_correlate_or_convolve` footprint of a kernel: offsets
    (ntaps, ndim) int64 and weights (ntaps,) float64 of the taps NI_Correlate
    visits, in its order (|w| <= DBL_EPSILON dropped)."""
    weights = np.asarray(weights, dtype=np.float64)
    ndim = weights.ndim
    origins = [int(origin)] * ndim if np.isscalar(origin) else [int(o) for o in origin]
    if len(origins) != ndim:
        raise ValueError('origin must have one entry per kernel axis')
    if convolution:
        weights = weights[tuple([slice(None, None, -1)] * ndim)]
        for ii in range(ndim):
            origins[ii] = -origins[ii]
            if not weights.shape[ii] & 1:
                origins[ii] -= 1
    for o, lenw in zip(origins, weights.shape):
        if (lenw // 2 + o < 0) or (lenw // 2 + o >= lenw):
            raise ValueError('invalid origin')
    keep = np.abs(weights) > np.finfo(np.float64).eps
    idx = np.argwhere(keep)                                  # C order
    centre = np.array([weights.shape[d] // 2 + origins[d] for d in range(ndim)])
    offs = (idx - centre).astype(np.int64).reshape(-1, ndim)
    return offs, np.ascontiguousarray(weights[keep], np.float64)


def _may_overlap(a, b):
    """Conservative test: the address ranges the two tensors span intersect."""
    if a.device != b.device or a.numel() == 0 or b.numel() == 0:
        return False

    def span(t):
        lo = t.data_ptr()
        hi = lo + (sum((n - 1) * abs(st) for n, st in zip(t.shape, t.stride())) + 1) * t.element_size()
        return lo, hi
    (a0, a1), (b0, b1) = span(a), span(b)
    return a0 < b1 and b0 < a1


def correlate_footprint(inp, out, offsets, weights, mode='reflect', cval=0.0):
    """out[i] = (T) sum_t weights[t] * inp[extend(i + offsets[t])], double
    accumulation in tap order.  inp/out: CUDA tensors, same shape, ndim <= 4."""
    _require_cuda(inp, 'inp')
    _require_cuda(out, 'out')
    if inp.shape != out.shape or inp.dtype != out.dtype or inp.device != out.device:
        raise ValueError('inp and out must share shape, dtype and device')
    if inp.dim() > 4:
        raise NotImplementedError('nd_amd_correlate handles up to 4 dimensions')
    if mode not in _lib.MODES:
        raise RuntimeError('boundary mode not supported')
    offsets = np.asarray(offsets, np.int64).reshape(-1, inp.dim())
    weights = np.ascontiguousarray(weights, np.float64)
    ntaps = len(weights)
    pad = 4 - inp.dim()
    dims = (1,) * pad + tuple(inp.shape)
    si = (0,) * pad + tuple(inp.stride())
    so = (0,) * pad + tuple(out.stride())
    offs4 = np.zeros((ntaps, 4), np.int64)
    offs4[:, pad:] = offsets
    dev = inp.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        taps_dev = None
        if ntaps > 128:
            taps_dev = torch.empty(24 * ntaps, dtype=torch.uint8, device=dev)
        _lib.check(L.nd_amd_correlate(
            _ptr(inp), _ptr(out), _DT[inp.dtype], _lib.i64_array(dims), _lib.i64_array(si),
            _lib.i64_array(so), ntaps,
            offs4.ctypes.data_as(C.POINTER(C.c_int64)),
            weights.ctypes.data_as(C.POINTER(C.c_double)),
            _lib.MODES[mode], float(cval), _ptr(taps_dev),
            0 if taps_dev is None else taps_dev.numel(), _stream_ptr(dev)))
    return out


def convolve(inp, kernel, out=None, mode='reflect', cval=0.0, origin=0):
    """scipy.ndimage.convolve(inp, kernel, output=out, mode, cval, origin) on a
    real CUDA tensor; kernel.ndim must equal inp.dim().

    A window over the last three axes of an x-contiguous array runs in the LDS-tiled kernel where the
    array lies.  When the array is laid out differently -- the reference's datasets are (y, x, time)
    with time fastest -- it is transposed on the device first and the result transposed back; the tap
    order over the window axes, hence the result, is unchanged.
    """
    kernel = np.asarray(kernel, np.float64)
    if kernel.ndim != inp.dim():
        raise RuntimeError('filter weights array has incorrect shape.')
    if out is None:
        out = torch.empty_like(inp)
    elif _may_overlap(inp, out):
        # scipy filters into a temporary when input and output share memory; the kernels read
        # neighbours of what they write
        out.copy_(convolve(inp, kernel, None, mode, cval, origin))
        return out
    nd = inp.dim()
    span = [d for d in range(nd) if kernel.shape[d] > 1]
    tail = list(range(nd - len(span), nd))
    # (round 5) the tiled kernel walks windows over the last THREE axes of x-contiguous arrays where they
    # lie -- (time, y, x) kernels, and windows along time or y alone, which used to be transposed to the
    # back first (a 3 x 1 x 1 window on 8 x 2048^2: 0.95 ms of transposes around 0.08 ms of filtering).  The
    # reference's (y, x, time) layout with a (y, x) window keeps its transpose kernels: its last axis is the
    # short one.
    # (only where the LAST axis is long -- it is the tiles' x axis; on time-fastest data, (y, x, time)
    # or (var, y, x, time) with a window over y and / or x, wide tiles would idle over a short time axis)
    in_place = (inp.stride(-1) == 1 and out.stride(-1) == 1 and all(d >= nd - 3 for d in span)
                and not (nd == 3 and span == [0, 1]) and inp.shape[-1] >= 64)
    if (0 < len(span) <= 2 and nd <= 4 and not in_place
            and (span != tail or inp.stride(-1) != 1 or out.stride(-1) != 1)
            and inp.numel() >= (1 << 16)):
        perm = [d for d in range(nd) if d not in span] + span
        origins = [int(origin)] * nd if np.isscalar(origin) else [int(o) for o in origin]
        k = np.transpose(kernel, perm)
        offs, w = footprint(k, [origins[d] for d in perm], convolution=True)
        # the reference's (y, x, time) layout with a (y, x) window: dedicated transpose kernels
        t = None
        if nd == 3 and span == [0, 1]:
            t = torch.empty((inp.shape[2], inp.shape[0], inp.shape[1]), dtype=inp.dtype, device=inp.device)
            if not relayout_planar(inp, t):
                t = None
        if t is None:
            t = inp.permute(*perm).contiguous()
        o = torch.empty_like(t)
        correlate_footprint(t, o, offs, w, mode, cval)
        if not (nd == 3 and span == [0, 1] and relayout_pixel_major(o, out)):
            inv = [perm.index(d) for d in range(nd)]
            out.copy_(o.permute(*inv))
        return out
    offs, w = footprint(kernel, origin, convolution=True)
    return correlate_footprint(inp, out, offs, w, mode, cval)


def correlate1d(inp, weights, axis, out, mode='reflect', cval=0.0):
    """scipy.ndimage.correlate1d(inp, weights, axis, out, mode, cval, origin=0) on a real CUDA
    tensor (ndim <= 4); `out` must not alias `inp`."""
    _require_cuda(inp, 'inp')
    _require_cuda(out, 'out')
    if inp.shape != out.shape or inp.dtype != out.dtype or inp.device != out.device:
        raise ValueError('inp and out must share shape, dtype and device')
    if inp.dim() > 4:
        raise NotImplementedError('nd_amd_correlate1d handles up to 4 dimensions')
    if mode not in _lib.MODES:
        raise RuntimeError('boundary mode not supported')
    weights = np.ascontiguousarray(weights, np.float64)
    pad = 4 - inp.dim()
    axis = axis % inp.dim()
    dev = inp.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nd_amd_correlate1d(
            _ptr(inp), _ptr(out), _DT[inp.dtype], _lib.i64_array((1,) * pad + tuple(inp.shape)),
            _lib.i64_array((0,) * pad + tuple(inp.stride())),
            _lib.i64_array((0,) * pad + tuple(out.stride())), axis + pad, len(weights),
            weights.ctypes.data_as(C.POINTER(C.c_double)), _lib.MODES[mode], float(cval),
            _stream_ptr(dev)))
    return out


def _gaussian_kernel1d(sigma, radius):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, order=0, radius)."""
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    return phi_x / phi_x.sum()


def _gaussian_yx_fused(inp, out, axes, nd, mode, truncate):
    """The common GaussianFilter(dims=('y', 'x')) case on x-contiguous float32 / float64 planes: both
    passes in one kernel (nd_amd_correlate1d_yx), the intermediate in the array dtype like scipy's.
    Returns False when the request is not one the fused kernel takes."""
    if (len(axes) != 2 or [ax for ax, _ in axes] != [nd - 2, nd - 1] or nd > 4
            or inp.dtype not in (torch.float32, torch.float64) or mode not in _lib.MODES or mode == 'constant'
            or inp.stride(-1) != 1 or out.stride(-1) != 1 or inp.data_ptr() == out.data_ptr()
            or inp.shape != out.shape or inp.device != out.device):
        return False
    lws = [int(truncate * sd + 0.5) for _, sd in axes]
    if lws[0] != lws[1] or lws[0] < 1:
        return False
    wy, wx = (np.ascontiguousarray(_gaussian_kernel1d(sd, lw)[::-1], np.float64)
              for (_, sd), lw in zip(axes, lws))
    pad = 4 - nd
    dev = inp.device
    with torch.cuda.device(dev):
        rc = _lib.lib().nd_amd_correlate1d_yx(
            _ptr(inp), _ptr(out), _DT[inp.dtype], _lib.i64_array((1,) * pad + tuple(inp.shape)),
            _lib.i64_array((0,) * pad + tuple(inp.stride())),
            _lib.i64_array((0,) * pad + tuple(out.stride())), len(wy),
            wy.ctypes.data_as(C.POINTER(C.c_double)), wx.ctypes.data_as(C.POINTER(C.c_double)),
            _lib.MODES[mode], _stream_ptr(dev))
    if rc == _lib.EUNSUPPORTED:
        return False
    _lib.check(rc)
    return True


def gaussian_filter(inp, sigma, out=None, mode='reflect', cval=0.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter(inp, sigma, output=out, mode, cval, truncate): one
    correlate1d pass per axis with sigma > 1e-15, each pass reading the previous pass's result
    in the array dtype (scipy filters `output` in place from the second axis on)."""
    if out is None:
        out = torch.empty_like(inp)
    elif _may_overlap(inp, out):
        out.copy_(gaussian_filter(inp, sigma, None, mode, cval, truncate))
        return out
    nd = inp.dim()
    sigmas = [float(sigma)] * nd if np.isscalar(sigma) else [float(s) for s in sigma]
    if len(sigmas) != nd:
        raise RuntimeError('sequence argument must have length equal to input rank')
    axes = [(ax, sd) for ax, sd in enumerate(sigmas) if sd > 1e-15]
    if not axes:
        out.copy_(inp)
        return out
    # The real / imaginary halves of a complex variable (and other views whose last axis is not
    # contiguous) would run in the per-element kernel: one packed copy in, one strided copy out
    # is cheaper by far (GaussianFilter on a (time, y, x) dataset with a complex64 C12: 6.8 -> 3.9 ms).
    if nd >= 1 and inp.numel() >= (1 << 16) and (inp.stride(-1) != 1 or out.stride(-1) != 1):
        packed = gaussian_filter(inp.contiguous(), sigma, None, mode, cval, truncate)
        out.copy_(packed)
        return out
    src = inp
    bufs = [out, None]
    if _gaussian_yx_fused(inp, out, axes, nd, mode, truncate):
        return out
    for n_done, (ax, sd) in enumerate(axes):
        lw = int(truncate * sd + 0.5)
        weights = _gaussian_kernel1d(sd, lw)[::-1]
        # the last pass must land in `out`; alternate between `out` and one scratch tensor
        remaining = len(axes) - n_done
        dst = out if remaining % 2 == 1 else None
        if dst is None:
            if bufs[1] is None:
                bufs[1] = torch.empty_like(out)
            dst = bufs[1]
        correlate1d(src, weights, ax, dst, mode, cval)
        src = dst
    return out


# ---------------------------------------------------------------------------
# non-local means
# ---------------------------------------------------------------------------
def raise_if_no_solution(status):
    """The deferred half of pixelwise_nlmeans_3d(..., status=t): reads the device flag (this is the
    host synchronisation) and raises the reference's ValueError('No solution') if it is set."""
    if status is not None and int(status.item()) != 0:
        raise ValueError('No solution')


def pixelwise_nlmeans_3d(arr, output, r, f, sigma, h, n_eff=-1, patch_mode=0,
                         neff_policy=1, global_shape=None, tile_offset=None,
                         core=None, status=None):
    """In-place into `output` like the reference.  arr/output: CUDA tensors
    (N0, N1, N2, nvars), any strides.

    patch_mode 0 = bit-compatible with the compiled reference (LP64: patch
    loops empty when any f > 0); 1 = true patch distances.
    neff_policy 1 raises ValueError('No solution') like a current build of the
    reference, 0 gives the self weight 0 of the shipped C.
    global_shape/tile_offset/core describe a halo-carrying tile (multi-GPU).
    status: an int32 device tensor of one element.  Without it (the reference's behaviour) a call
    with n_eff >= 0 reads the find_weight flag back and raises at once -- one host
    synchronisation per call.  With it the library writes the flag there (zeroing it first) and
    the call stays asynchronous: a pipeline step then has no host stall, and the caller checks
    with raise_if_no_solution(status) when it next needs the host anyway.
    """
    _require_cuda(arr, 'arr')
    _require_cuda(output, 'output')
    if arr.dim() != 4 or output.shape != arr.shape or output.dtype != arr.dtype:
        raise ValueError('arr and output must be 4-D (N0, N1, N2, nvars) and alike')
    r = [int(v) for v in r]
    f = [int(v) for v in f]
    if len(r) != 3 or len(f) != 3:
        raise ValueError('r and f must have three entries')
    N = tuple(arr.shape[:3])
    G = tuple(global_shape) if global_shape is not None else N
    toff = tuple(tile_offset) if tile_offset is not None else (0, 0, 0)
    clo = tuple(c[0] for c in core) if core is not None else (0, 0, 0)
    chi = tuple(c[1] for c in core) if core is not None else N
    dev = arr.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        deferred = status is not None
        if deferred and (status.dtype != torch.int32 or status.numel() != 1 or status.device != dev):
            raise ValueError('status must be one int32 element on the device of arr')
        if not deferred and n_eff >= 0 and neff_policy == 1:
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.nd_amd_nlmeans3d(
            _ptr(arr), _ptr(output), _DT[arr.dtype], _lib.i64_array(N), arr.shape[3],
            _lib.i64_array(arr.stride()), _lib.i64_array(output.stride()),
            _lib.u32_array(r), _lib.u32_array(f), float(sigma), float(h), float(n_eff),
            int(patch_mode), int(neff_policy), _ptr(status),
            _lib.i64_array(G), _lib.i64_array(toff), _lib.i64_array(clo),
            _lib.i64_array(chi), _stream_ptr(dev)))
        if not deferred:
            raise_if_no_solution(status)
    return output
