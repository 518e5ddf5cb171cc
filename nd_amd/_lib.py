"""
nd_amd/_lib.py -- ctypes binding of libnd_amd.so (include/nd_amd.h).

The HIP library is the only compute path of this package.  If it is missing the
import of any op raises; there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ND_AMD_LIB points the loader at an alternative build of the same library (tuning experiments)
LIB_PATH = os.environ.get('ND_AMD_LIB') or os.path.join(_HERE, 'libnd_amd.so')

F32, F64 = 0, 1
OK, EINVAL, EWORKSPACE, EHIP, ENOSOLUTION, EUNSUPPORTED = 0, -1, -2, -3, -4, -5
MODES = {'reflect': 0, 'constant': 1, 'nearest': 2, 'mirror': 3, 'wrap': 4,
         'grid-constant': 1, 'grid-mirror': 0, 'grid-wrap': 4}
KERNEL_NAMES = {1: 'omnibus_c2_global', 2: 'omnibus_c2_search', 3: 'correlate',
                4: 'nlmeans', 5: 'boxcar_tiled', 6: 'nlmeans_tiled', 7: 'correlate1d',
                8: 'relayout', 9: 'omnibus_c2_dense', 10: 'omnibus_c2_fused', 11: 'omnibus_c2_sample', 12: 'omnibus_c2_exact'}

# every symbol include/nd_amd.h declares
SYMBOLS = ('nd_amd_abi_version', 'nd_amd_last_error',
           'nd_amd_omnibus_c2_workspace_bytes', 'nd_amd_omnibus_c2',
           'nd_amd_omnibus_c2_pixel_major',
           'nd_amd_omnibus_c2_ml_workspace_bytes', 'nd_amd_omnibus_c2_ml',
           'nd_amd_omnibus_c3_workspace_bytes', 'nd_amd_omnibus_c3', 'nd_amd_omnibus_c3_pixel_major',
           'nd_amd_correlate', 'nd_amd_correlate1d', 'nd_amd_correlate1d_yx', 'nd_amd_nlmeans3d',
           'nd_amd_relayout_planar', 'nd_amd_relayout_planar_complex',
           'nd_amd_relayout_pixel_major', 'nd_amd_split_complex', 'nd_amd_merge_complex',
           'nd_amd_timing_enable', 'nd_amd_timing_collect', 'nd_amd_timing_dropped',
           'nd_amd_timing_select')

_lib = None


class NdAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libnd_amd error %d: %s' % (code, msg))
        self.code = code


def lib():
    """Load libnd_amd.so (built by `python -m nd_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'nd_amd: %s is missing -- build it with `python -m nd_amd.build` '
            '(hipcc, gfx950).  There is no fallback path.' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    i64, dbl, vp, i32 = C.c_int64, C.c_double, C.c_void_p, C.c_int
    L.nd_amd_abi_version.restype = i32
    L.nd_amd_abi_version.argtypes = []
    L.nd_amd_last_error.restype = C.c_char_p
    L.nd_amd_last_error.argtypes = []
    L.nd_amd_omnibus_c2_workspace_bytes.restype = C.c_size_t
    L.nd_amd_omnibus_c2_workspace_bytes.argtypes = [i32, i64, i64, i64,
                                                    C.POINTER(C.c_size_t)]
    L.nd_amd_omnibus_c2.restype = i32
    L.nd_amd_omnibus_c2.argtypes = ([vp] * 4 + [i32] + [i64] * 6 + [C.c_uint32, dbl]
                                    + [vp, vp, vp, vp, C.c_size_t, vp])
    L.nd_amd_omnibus_c2_ml_workspace_bytes.restype = C.c_size_t
    L.nd_amd_omnibus_c2_ml_workspace_bytes.argtypes = [i32, i64, i64, i64, i32]
    L.nd_amd_omnibus_c2_ml.restype = i32
    L.nd_amd_omnibus_c2_ml.argtypes = ([vp] * 4 + [i32] + [i64] * 6 + [i32, dbl]
                                       + [vp, vp, vp, vp, C.c_size_t, vp])
    L.nd_amd_omnibus_c2_pixel_major.restype = i32
    L.nd_amd_omnibus_c2_pixel_major.argtypes = ([vp] * 4 + [i32] + [i64] * 3 + [C.POINTER(i64)]
                                                + [C.c_uint32, dbl] + [vp, vp, vp, vp, C.c_size_t, vp])
    L.nd_amd_omnibus_c3_workspace_bytes.restype = C.c_size_t
    L.nd_amd_omnibus_c3_workspace_bytes.argtypes = [i64, i64, i64]
    L.nd_amd_omnibus_c3.restype = i32
    L.nd_amd_omnibus_c3.argtypes = ([C.POINTER(vp), i32] + [i64] * 6 + [C.c_uint32, dbl]
                                    + [vp, vp, vp, vp, C.c_size_t, vp])
    L.nd_amd_omnibus_c3_pixel_major.restype = i32
    L.nd_amd_omnibus_c3_pixel_major.argtypes = ([C.POINTER(vp), i32] + [i64] * 3 + [C.POINTER(i64)]
                                                + [C.c_uint32, dbl] + [vp, vp, vp, vp, C.c_size_t, vp])
    L.nd_amd_correlate.restype = i32
    L.nd_amd_correlate.argtypes = [vp, vp, i32, C.POINTER(i64), C.POINTER(i64),
                                   C.POINTER(i64), i64, C.POINTER(i64),
                                   C.POINTER(dbl), i32, dbl, vp, C.c_size_t, vp]
    L.nd_amd_correlate1d.restype = i32
    L.nd_amd_correlate1d.argtypes = [vp, vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                     i32, i32, C.POINTER(dbl), i32, dbl, vp]
    L.nd_amd_merge_complex.restype = i32
    L.nd_amd_merge_complex.argtypes = [vp, vp, vp, i32, i64, vp]
    L.nd_amd_split_complex.restype = i32
    L.nd_amd_split_complex.argtypes = [vp, vp, vp, i32, i64, vp]
    L.nd_amd_correlate1d_yx.restype = i32
    L.nd_amd_correlate1d_yx.argtypes = [vp, vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                        i32, C.POINTER(dbl), C.POINTER(dbl), i32, vp]
    L.nd_amd_nlmeans3d.restype = i32
    L.nd_amd_nlmeans3d.argtypes = [vp, vp, i32, C.POINTER(i64), i64, C.POINTER(i64),
                                   C.POINTER(i64), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), dbl, dbl, dbl, i32, i32, vp,
                                   C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                   C.POINTER(i64), vp]
    L.nd_amd_relayout_planar.restype = i32
    L.nd_amd_relayout_planar.argtypes = [vp, vp, i32, i64, i64, i64, i64, vp]
    L.nd_amd_relayout_planar_complex.restype = i32
    L.nd_amd_relayout_planar_complex.argtypes = [vp, vp, vp, i32, i64, i64, i64, vp]
    L.nd_amd_relayout_pixel_major.restype = i32
    L.nd_amd_relayout_pixel_major.argtypes = [vp, vp, i32, i64, i64, i64, i64, vp]
    L.nd_amd_timing_enable.restype = i32
    L.nd_amd_timing_enable.argtypes = [i32]
    L.nd_amd_timing_collect.restype = i32
    L.nd_amd_timing_collect.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_float), i32,
                                        C.POINTER(i32)]
    L.nd_amd_timing_dropped.restype = i32
    L.nd_amd_timing_dropped.argtypes = []
    L.nd_amd_timing_select.restype = i32
    L.nd_amd_timing_select.argtypes = [C.c_uint64]
    v = L.nd_amd_abi_version()
    if v != 1:
        raise ImportError('nd_amd: libnd_amd.so has ABI version %d, expected 1' % v)
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise NdAmdError(rc, lib().nd_amd_last_error().decode('utf-8', 'replace'))


def i64_array(values):
    return (C.c_int64 * len(values))(*[int(v) for v in values])


def u32_array(values):
    return (C.c_uint32 * len(values))(*[int(v) for v in values])


def timing_enable(capacity):
    check(lib().nd_amd_timing_enable(int(capacity)))


def timing_select(names=None):
    """time only the kernels named (KERNEL_NAMES values); None = all.  Reset by timing_enable."""
    mask = 0
    if names:
        ids = {v: k for k, v in KERNEL_NAMES.items()}
        for n in names:
            mask |= 1 << ids[n]
    check(lib().nd_amd_timing_select(mask))


def timing_dropped():
    """launches that found the timing ring full since the last call (0 = every launch was timed)."""
    return int(lib().nd_amd_timing_dropped())


def timing_collect(max_n=65536):
    """-> list of (kernel name, milliseconds) in launch order; synchronises."""
    ids = (C.c_int32 * max_n)()
    ms = (C.c_float * max_n)()
    n = C.c_int(0)
    check(lib().nd_amd_timing_collect(ids, ms, max_n, C.byref(n)))
    return [(KERNEL_NAMES.get(ids[i], str(ids[i])), float(ms[i])) for i in range(n.value)]
