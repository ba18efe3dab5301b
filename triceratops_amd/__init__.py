"""triceratops_amd: MI355X-native marginal-likelihood hot path of triceratops.

Mirrors the reference's module layout for this path only:
  _numerics, likelihoods, marginal_likelihoods, priors, funcs, triceratops.
Compute runs in hand-written HIP kernels behind the C ABI of include/trx.h (libtrx.so).
"""
__version__ = "0.1.0"
