"""Numerically stable reductions used by the marginal likelihoods.

Same call surface as the reference's triceratops/_numerics.py:
  _log_mean_exp(logw, *, N_total)   (_numerics.py:12-51)  -> HIP kernel trx_log_mean_exp
  _normalize_probabilities(lnZ)     (_numerics.py:54-76)  -> host arithmetic over the 18-75
                                                             scenario evidences
"""
import numpy as np
import torch

from . import _lib


def _log_mean_exp(logw, *, N_total: int) -> float:
    """log(mean(exp(logw))) over N_total draws, evaluated on the GPU.

    -inf and NaN entries carry zero weight but count in the denominator; any +inf gives +inf;
    no finite entry gives -inf; N_total must equal len(logw) (ValueError otherwise, exactly like
    the reference guard at _numerics.py:40-45).  `logw` may be a numpy array or a CUDA tensor.
    """
    size = logw.numel() if isinstance(logw, torch.Tensor) else np.size(logw)
    if N_total != size:
        raise ValueError(
            f"N_total ({N_total}) must equal len(logw) ({size}). "
            "Passing len(lnL[finite]) instead of len(lnL) would silently "
            "overestimate evidence for scenarios with geometric exclusions."
        )
    d = _lib.dev(logw).reshape(-1)
    return float(_lib.log_mean_exp(d, N_total).cpu()[0])


def _normalize_probabilities(lnZ):
    """Scenario probabilities exp(lnZ - logsumexp(lnZ)) and a status string:
    'ok', 'all_neginf' (every evidence is -inf) or 'anomaly' (a NaN or +inf is present);
    the two degenerate cases return all-zero probabilities."""
    lnZ = np.asarray(lnZ, dtype=np.float64)
    if np.any(np.isnan(lnZ)) or np.any(np.isposinf(lnZ)):
        return np.zeros(len(lnZ)), 'anomaly'
    if np.all(np.isneginf(lnZ)):
        return np.zeros(len(lnZ)), 'all_neginf'
    top = np.max(lnZ)
    with np.errstate(divide="ignore"):
        lse = np.log(np.sum(np.exp(lnZ - top))) + top
    return np.exp(lnZ - lse), 'ok'
