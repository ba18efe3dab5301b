"""Random-number sources and the TRILEGAL population of the device sampling modes.

set_sampling("device") and "numpy-device" run a scenario's per-draw chain in ONE HIP kernel (fused.py,
csrc/trx_draw.hip).  What that kernel is fed lives here:
  TorchRng        set_sampling("device"): the kernel draws its own Philox numbers (fused.PHILOX); with
                  fused.PHILOX = False torch's device generator fills staged arrays in the reference's draw order
  NumpyStreamRng  set_sampling("numpy-device"): numpy's global legacy stream, consumed exactly where the
                  reference consumes it (including the draws it never uses), copied to the device
  _Field          the TRILEGAL population of the D / B scenarios as device columns
The elementwise torch expression of the same chain (round 1's device path, ~360 launches per scenario) is
test infrastructure now: tests/torch_pipeline.py, the cross-check of the draw kernel.
"""
import numpy as np
import torch

from . import _lib
from . import marginal_likelihoods as ml

F64 = torch.float64


class TorchRng:
    """random numbers from torch's device generator (Philox), in the reference's draw order.
    `discard` marks the places where the reference draws uniforms it never uses (the argument
    of sample_ecc, lnZ_BEB's companion mass ratios): nothing is generated for them here."""

    def uniform(self, n, device):
        return torch.rand(n, dtype=F64, device=device)

    def beta(self, n, a, b, device):
        # X / (X + Y) with X ~ Gamma(a), Y ~ Gamma(b): what torch.distributions.Beta computes
        # through a Dirichlet, without building distribution objects on every call
        x = torch._standard_gamma(torch.full((n,), a, dtype=F64, device=device))
        y = torch._standard_gamma(torch.full((n,), b, dtype=F64, device=device))
        return x / (x + y)

    def randint(self, hi, n, device):
        return torch.randint(0, hi, (n,), device=device)

    def discard(self, n):
        pass


class NumpyStreamRng:
    """numpy's global legacy stream, consumed exactly where the reference consumes it (including
    the draws it never uses), copied to the device: the same Monte-Carlo draws as the reference
    under the same np.random.seed, with everything downstream of the uniforms on the GPU."""

    def uniform(self, n, device):
        return torch.as_tensor(np.random.rand(n)).to(device)

    def beta(self, n, a, b, device):
        return torch.as_tensor(np.random.beta(a, b, size=n)).to(device)

    def randint(self, hi, n, device):
        return torch.as_tensor(np.random.randint(0, hi, n)).to(device)

    def discard(self, n):
        np.random.rand(n)


RNG = TorchRng()


class _Field:
    """TRILEGAL population on the device (built from the host reader, a few thousand stars)"""

    def __init__(self, ctx, trilegal_fname, Tmag, Jmag, Hmag, Kmag, mission, need_ldc):
        h = ml._Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
        d = ctx["device"]
        self.N_comp = h.N_comp
        self.masses, self.loggs, self.Teffs = (_lib.dev(v, d) for v in (h.masses, h.loggs, h.Teffs))
        self.fluxratios = _lib.dev(h.fluxratios, d)
        self.delta = {k: _lib.dev(v, d) for k, v in (("T", h.dT), ("J", h.dJ), ("H", h.dH), ("K", h.dK))}
        self.radii = _lib.dev(h.radii(), d)
        if need_ldc:
            u1, u2 = ml._ldc(mission).field_stars(h.Teffs, h.loggs, h.Zs)
            self.u1, self.u2 = _lib.dev(u1, d), _lib.dev(u2, d)

    def band_delta(self, filt):
        return self.delta.get(filt, self.delta["T"])

    def band_fluxratio(self, filt):
        dm = self.band_delta(filt)
        return 10 ** (dm / 2.5) / (1 + 10 ** (dm / 2.5))
