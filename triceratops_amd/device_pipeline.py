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
        h = self._host = ml._Field(trilegal_fname, Tmag, Jmag, Hmag, Kmag)
        d = self._device = ctx["device"]
        self._mission = mission
        self.N_comp = h.N_comp
        # the nine columns and the four band flux ratios in ONE upload (a few thousand stars): rows of one block.
        # (the band flux ratios were torch expressions on the device until round 3: three launches and a stream
        # synchronisation per star for numbers numpy has in microseconds -- marginal_likelihoods._Field)
        bands = [h.band_fluxratio(b) for b in ("T", "J", "H", "K")]
        block = _lib.dev(np.stack([h.masses, h.loggs, h.Teffs, h.fluxratios, h.dT, h.dJ, h.dH, h.dK, h.radii()] + bands), d)
        self.masses, self.loggs, self.Teffs, self.fluxratios = block[0], block[1], block[2], block[3]
        self.delta = {"T": block[4], "J": block[5], "H": block[6], "K": block[7]}
        self.radii = block[8]
        self._band_fr = {"T": block[9], "J": block[10], "H": block[11], "K": block[12]}
        self.u1 = self.u2 = None
        if need_ldc:
            self.need_ldc()

    def need_ldc(self):
        """per-star limb-darkening coefficients (the B scenarios; the D scenarios never ask)"""
        if self.u1 is None:
            h = self._host
            u1, u2 = ml._ldc(self._mission).field_stars(h.Teffs, h.loggs, h.Zs)
            both = _lib.dev(np.stack([u1, u2]), self._device)
            self.u1, self.u2 = both[0], both[1]

    def band_delta(self, filt):
        return self.delta.get(filt, self.delta["T"])

    def band_fluxratio(self, filt):
        return self._band_fr.get(filt, self._band_fr["T"])
