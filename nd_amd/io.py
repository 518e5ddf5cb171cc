"""
nd_amd/io.py -- the naming convention that carries complex covariance terms through real-valued
storage, as the reference defines it (nd/io.py:26-123): a complex variable `C12` is stored as the
pair `C12__re` / `C12__im`; on the way back `C12_real` / `C12_imag` are recognised as well.
Works on xarray and nd_amd.xr_lite datasets, host arrays or device tensors (the split of a device
tensor is two strided views of the interleaved complex64 memory -- no copy).
"""
from . import _adapter

SPLIT_SUFFIXES = ('__re', '__im')
_REAL_SUFFIXES = ('__re', '_real')
_IMAG_SUFFIXES = ('__im', '_imag')


def _as_dataset(ds):
    ns = _adapter.namespace(ds)
    if isinstance(ds, ns.DataArray):
        return ds.to_dataset(name=ds.name if ds.name is not None else 'data')
    return ds


def disassemble_complex(ds, inplace=False):
    """Replace every complex variable `v` by `v__re` and `v__im`.  Returns the new dataset, or None
    when `inplace` (same contract as nd/io.py:26-70)."""
    ds = _as_dataset(ds)
    target = ds if inplace else ds.copy()
    complex_names = [name for name in list(ds.data_vars)
                     if _adapter.iscomplexobj(ds[name].values)]
    for name in complex_names:
        var = ds[name]
        target[name + SPLIT_SUFFIXES[0]] = var.real
        target[name + SPLIT_SUFFIXES[1]] = var.imag
        del target[name]
    return None if inplace else target


def _split_name(name):
    """('C12', 're') for 'C12__re' / 'C12_real', ('C12', 'im') for the imaginary forms, else None.
    The longest stem wins when more than one suffix matches, like the reference's greedy regex."""
    for part, suffixes in (('re', _REAL_SUFFIXES), ('im', _IMAG_SUFFIXES)):
        for suffix in sorted(suffixes, key=len):
            if name.endswith(suffix):
                return name[:len(name) - len(suffix)], part
    return None


def assemble_complex(ds, inplace=False):
    """Inverse of disassemble_complex: every stem that has both a real and an imaginary part becomes
    one complex variable again; unpaired parts are left alone (nd/io.py:73-123)."""
    target = ds if inplace else ds.copy()
    parts = {}
    for name in list(ds.data_vars):
        hit = _split_name(name)
        if hit is not None:
            parts.setdefault(hit[0], {}).setdefault(hit[1], name)
    for stem, found in parts.items():
        if 're' in found and 'im' in found:
            target[stem] = target[found['re']] + target[found['im']] * 1j
            del target[found['re']]
            del target[found['im']]
    return None if inplace else target
