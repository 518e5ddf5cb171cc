"""
nd_amd/io.py -- complex <-> (real, imag) variable convention of the reference
(nd/io.py:26-123): `C12` (complex) <-> `C12__re`, `C12__im`; `_real`/`_imag` are accepted on
reassembly like in the reference.  Works on xarray and nd_amd.xr_lite datasets.
"""
import re

from . import _adapter


def disassemble_complex(ds, inplace=False):
    ns = _adapter.namespace(ds)
    if isinstance(ds, ns.DataArray):
        name = ds.name
        if name is None:
            name = 'data'
        ds = ds.to_dataset(name=name)
    new_ds = ds if inplace else ds.copy()
    for vn in list(ds.data_vars):
        var = ds[vn]
        if not _adapter.iscomplexobj(var.values):
            continue
        new_ds[vn + '__re'] = var.real
        new_ds[vn + '__im'] = var.imag
        del new_ds[vn]
    if not inplace:
        return new_ds


def assemble_complex(ds, inplace=False):
    new_ds = ds if inplace else ds.copy()
    endings = {'re': ['_real', '__re'], 'im': ['_imag', '__im']}
    matches = {}
    for part, end in endings.items():
        rex = re.compile('(?P<stem>.*)(?:{})'.format('|'.join(end)))
        found = [rex.match(vn) for vn in ds.data_vars]
        matches[part] = [m for m in found if m is not None]
    stems = set(m.group('stem') for m in matches['re'] + matches['im'])
    for vn in stems:
        vn_re = next((m for m in matches['re'] if m.group(1) == vn), None)
        vn_im = next((m for m in matches['im'] if m.group(1) == vn), None)
        if vn_re is not None and vn_im is not None:
            new_ds[vn] = new_ds[vn_re.group(0)] + new_ds[vn_im.group(0)] * 1j
            del new_ds[vn_re.group(0)]
            del new_ds[vn_im.group(0)]
    if not inplace:
        return new_ds
