"""
nd_amd -- MI355X (gfx950) implementation of the per-pixel compute path of jnhansen/nd:
the OmnibusTest complex-Wishart change detector and the windowed filters (non-local means,
boxcar, kernel convolution) behind the reference's `Algorithm.apply(dataset)` interface.

    from nd_amd.change import OmnibusTest          # nd.change.OmnibusTest
    from nd_amd.filters import NLMeansFilter, BoxcarFilter, ConvolutionFilter

The arithmetic runs in hand-written HIP kernels (nd_amd/csrc, built by `python -m nd_amd.build`
into nd_amd/libnd_amd.so) reached through a plain C ABI (include/nd_amd.h).  There is no CPU
fallback: importing an op without the library raises.
"""
__version__ = '0.1.0'
