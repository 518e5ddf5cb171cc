"""Seeded synthetic inputs shared by the CPU and GPU tests (numpy only)."""
import numpy as np


def wishart_c2(rng, shape, looks=9, dtype=np.float32, corr=0.3, power=(1.0, 0.5)):
    """n-look complex-Wishart dual-pol covariance samples.
    Returns c11, c12re, c12im, c22 arrays of `shape`."""
    def cn(sh):
        return (rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) / np.sqrt(2.0)
    s1 = cn((looks,) + tuple(shape))
    s2 = corr * s1 + np.sqrt(1 - corr ** 2) * cn((looks,) + tuple(shape))
    s1 = s1 * np.sqrt(power[0])
    s2 = s2 * np.sqrt(power[1])
    c11 = (np.abs(s1) ** 2).mean(axis=0)
    c22 = (np.abs(s2) ** 2).mean(axis=0)
    c12 = (s1 * np.conj(s2)).mean(axis=0)
    return (c11.astype(dtype), c12.real.astype(dtype), c12.imag.astype(dtype),
            c22.astype(dtype))


def omnibus_stack(seed, k, ny, nx, looks=9, dtype=np.float32, change_frac=0.05,
                  factor=4.0):
    """Planar (time, y, x) Wishart stack with a fraction of pixels stepping in
    power by `factor` at a random date.  Returns 4 arrays (k, ny, nx)."""
    rng = np.random.default_rng(seed)
    planes = wishart_c2(rng, (k, ny, nx), looks, np.float64)
    if change_frac > 0:
        mask = rng.random((ny, nx)) < change_frac
        t0 = rng.integers(1, max(k, 2), size=(ny, nx))
        step = (np.arange(k)[:, None, None] >= t0[None]) & mask[None]
        gain = np.where(step, factor, 1.0)
        # second population: a drop, so both directions and multi-change pixels occur
        mask2 = rng.random((ny, nx)) < change_frac / 2
        t1 = rng.integers(1, max(k, 2), size=(ny, nx))
        gain = gain * np.where((np.arange(k)[:, None, None] >= t1[None]) & mask2[None], 0.3, 1.0)
        planes = tuple(p * gain for p in planes)
    return tuple(np.ascontiguousarray(p.astype(dtype)) for p in planes)


def reference_test_dataset(dims, mean, sigma, seed=42,
                           var=('C11', 'C12__im', 'C12__re', 'C22')):
    """nd.testing.generate_test_dataset (nd/testing.py:34-70) restated for
    plain arrays: np.random.seed(seed), one float64 normal draw per variable in
    `var` order, shape = dims values."""
    np.random.seed(seed)
    if np.isscalar(mean):
        mean = [mean] * len(var)
    out = {}
    for v, m in zip(var, mean):
        out[v] = np.random.normal(m, sigma, tuple(dims.values()))
    return out


def lite_test_dataset(dims=None, var=('C11', 'C12__im', 'C12__re', 'C22'), mean=0, sigma=1,
                      seed=42):
    """nd.testing.generate_test_dataset (nd/testing.py:34-70) as an nd_amd.xr_lite.Dataset:
    same seed, same draw order, same values; coordinates reduced to plain index arrays."""
    from collections import OrderedDict
    from nd_amd import xr_lite
    if dims is None:
        dims = OrderedDict([('y', 20), ('x', 20), ('time', 10)])
    data = reference_test_dataset(dims, mean, sigma, seed, var)
    coords = OrderedDict()
    for name, size in dims.items():
        if name == 'y':
            coords[name] = np.linspace(60.0, 50.0, size)
        elif name == 'x':
            coords[name] = np.linspace(-10.0, 0.0, size)
        else:
            coords[name] = np.arange(size)
    ds = xr_lite.Dataset(coords=coords, attrs={'attr1': 1, 'attr2': 2, 'attr3': 3})
    for v in var:
        ds[v] = (tuple(dims.keys()), data[v])
    return ds


def wishart_c3(rng, shape, looks=9, dtype=np.float32):
    """n-look complex-Wishart full-pol samples: 9 real planes
    [C11, C22, C33, C12re, C12im, C13re, C13im, C23re, C23im] of `shape`."""
    def cn(sh):
        return (rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) / np.sqrt(2.0)
    s1 = cn((looks,) + tuple(shape))
    s2 = 0.4 * s1 + 0.9 * cn((looks,) + tuple(shape))
    s3 = 0.2 * s1 - 0.3j * s2 + 0.8 * cn((looks,) + tuple(shape))
    c = lambda a, b: (a * np.conj(b)).mean(axis=0)
    c12, c13, c23 = c(s1, s2), c(s1, s3), c(s2, s3)
    out = [c(s1, s1).real, c(s2, s2).real, c(s3, s3).real, c12.real, c12.imag, c13.real, c13.imag,
           c23.real, c23.imag]
    return [np.ascontiguousarray(o.astype(dtype)) for o in out]


def omnibus_stack_c3(seed, k, ny, nx, looks=9, dtype=np.float32, change_frac=0.1, factor=4.0):
    rng = np.random.default_rng(seed)
    planes = wishart_c3(rng, (k, ny, nx), looks, np.float64)
    if change_frac > 0:
        mask = rng.random((ny, nx)) < change_frac
        t0 = rng.integers(1, max(k, 2), size=(ny, nx))
        gain = np.where((np.arange(k)[:, None, None] >= t0[None]) & mask[None], factor, 1.0)
        planes = [p * gain for p in planes]
    return [np.ascontiguousarray(p.astype(dtype)) for p in planes]
