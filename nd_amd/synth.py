"""
nd_amd/synth.py -- synthetic SAR covariance stacks generated on the device
(bench.py and the large-size property tests).

n-look complex-Wishart dual-pol samples: s1, s2 ~ CN(0, 1) per look,
C11 = mean|s1|^2, C22 = mean|s2|^2, C12 = mean(s1 conj(s2)); a fraction of the
pixels gets a x`factor` power step at a random date.  Layout: one float tensor
(4, time, y, x), planes [C11, C12re, C12im, C22], x fastest.
"""
import torch


# The date planes of a device stack are padded by this many elements: with a power-of-two plane
# size (4096 x 4096 x 4 B = 64 MiB) the 96 streams a pixel block reads would otherwise start at
# addresses that differ only above bit 26, which costs ~7 % of HBM read bandwidth on MI355X
# (profiles/r01_probe_bandwidth.txt).
DATE_PAD = 64


def empty_stack(nvar, k, ny, nx, device, dtype=torch.float32, date_pad=DATE_PAD):
    """Uninitialised planar stack (nvar, time, y, x), x fastest, each date plane
    contiguous and `date_pad` elements apart from the next (16-byte aligned)."""
    npix = ny * nx
    pad = date_pad if npix % 1024 == 0 else 0
    buf = torch.empty((nvar, k, npix + pad), dtype=dtype, device=device)
    return buf[:, :, :npix].view(nvar, k, ny, nx)


def wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device='cuda', dtype=torch.float32,
                     change_frac=0.01, factor=4.0, corr=0.3, date_pad=DATE_PAD, out=None):
    """`out`: optional (4, k, ny, nx) tensor (e.g. the core view of a tiles.RowShard) to fill
    instead of allocating a stack."""
    dev = torch.device(device) if out is None else out.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    if out is None:
        out = empty_stack(4, k, ny, nx, dev, dtype, date_pad)
    elif tuple(out.shape) != (4, k, ny, nx):
        raise ValueError('out must have shape (4, k, ny, nx)')
    dtype = out.dtype
    if change_frac > 0:
        mask = torch.rand((ny, nx), generator=gen, device=dev) < change_frac
        t0 = torch.randint(1, max(k, 2), (ny, nx), generator=gen, device=dev)
    c = (1.0 - corr * corr) ** 0.5
    for t in range(k):
        c11 = torch.zeros((ny, nx), dtype=torch.float32, device=dev)
        c22 = torch.zeros_like(c11)
        c12r = torch.zeros_like(c11)
        c12i = torch.zeros_like(c11)
        for _ in range(looks):
            a = torch.randn((4, ny, nx), generator=gen, device=dev) * (0.5 ** 0.5)
            s1r, s1i = a[0], a[1]
            s2r = corr * s1r + c * a[2]
            s2i = corr * s1i + c * a[3]
            c11 += s1r * s1r + s1i * s1i
            c22 += s2r * s2r + s2i * s2i
            c12r += s1r * s2r + s1i * s2i          # s1 conj(s2)
            c12i += s1i * s2r - s1r * s2i
        planes = [c11, c12r, c12i, c22]
        for v in range(4):
            p = planes[v] / looks
            if change_frac > 0:
                p = torch.where(mask & (t0 <= t), p * factor, p)
            out[v, t] = p.to(dtype)
    return out


# plane order of nd_amd_omnibus_c3 (nd_amd/change.py:_VARS3)
C3_PLANES = ('C11', 'C22', 'C33', 'C12re', 'C12im', 'C13re', 'C13im', 'C23re', 'C23im')


def wishart_c3_stack(k, ny, nx, looks=9, seed=1234, device='cuda', dtype=torch.float32,
                     change_frac=0.01, factor=4.0, corr=0.3, date_pad=DATE_PAD, cycle=0):
    """n-look complex-Wishart full-pol samples, planar (9, time, y, x) in the order C3_PLANES:
    s1, s2, s3 ~ CN(0, 1) per look with s2, s3 correlated to s1 by `corr`; C_ab = mean(s_a conj(s_b)).
    A fraction of the pixels gets a x`factor` power step at a random date.
    cycle > 0: only the first `cycle` dates are drawn and the others repeat them (the step still
    applies per date) -- a tenth of the device launches, for runs under a profiler's counter pass,
    which gives up beyond some ten thousand dispatches."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    out = empty_stack(9, k, ny, nx, dev, dtype, date_pad)
    if change_frac > 0:
        mask = torch.rand((ny, nx), generator=gen, device=dev) < change_frac
        t0 = torch.randint(1, max(k, 2), (ny, nx), generator=gen, device=dev)
    c = (1.0 - corr * corr) ** 0.5
    drawn = {}
    for t in range(k):
        if cycle > 0 and t >= cycle:
            for v in range(9):
                p = drawn[(v, t % cycle)]
                if change_frac > 0:
                    p = torch.where(mask & (t0 <= t), p * factor, p)
                out[v, t] = p.to(dtype)
            continue
        acc = [torch.zeros((ny, nx), dtype=torch.float32, device=dev) for _ in range(9)]
        for _ in range(looks):
            a = torch.randn((6, ny, nx), generator=gen, device=dev) * (0.5 ** 0.5)
            s1r, s1i = a[0], a[1]
            s2r, s2i = corr * s1r + c * a[2], corr * s1i + c * a[3]
            s3r, s3i = corr * s1r + c * a[4], corr * s1i + c * a[5]
            acc[0] += s1r * s1r + s1i * s1i
            acc[1] += s2r * s2r + s2i * s2i
            acc[2] += s3r * s3r + s3i * s3i
            acc[3] += s1r * s2r + s1i * s2i          # s1 conj(s2)
            acc[4] += s1i * s2r - s1r * s2i
            acc[5] += s1r * s3r + s1i * s3i          # s1 conj(s3)
            acc[6] += s1i * s3r - s1r * s3i
            acc[7] += s2r * s3r + s2i * s3i          # s2 conj(s3)
            acc[8] += s2i * s3r - s2r * s3i
        for v in range(9):
            p = acc[v] / looks
            if cycle > 0:
                drawn[(v, t)] = p
            if change_frac > 0:
                p = torch.where(mask & (t0 <= t), p * factor, p)
            out[v, t] = p.to(dtype)
    return out
