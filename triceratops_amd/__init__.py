"""triceratops_amd: MI355X-native marginal-likelihood hot path of triceratops.

Mirrors the reference's module layout for this path only:
  _numerics, likelihoods, marginal_likelihoods, priors, funcs, triceratops.
Compute runs in hand-written HIP kernels behind the C ABI of include/trx.h (libtrx.so).
"""
__version__ = "0.2.0"

import os as _os

# The HIP runtime multiplexes a process's streams onto 4 hardware queues unless told otherwise, and two streams that
# share a queue serialise: calc_probs on three streams then runs slower than on two (64 TOIs: 0.223 s a step on two
# streams, 0.263 on three, 0.226 on four, 0.190 on six; with 8 queues 0.224 / 0.185 / 0.172 / 0.180 --
# profiles/r04/hw_queues.txt).  Read by the runtime when it initialises (the first HIP call of the process), so it
# has to be in the environment before that; a value the user has set is left alone.
# Sixteen since round 6: a live RCCL communicator takes hardware queues of its own, and with eight the worker streams of a
# rank then share -- a 64-target step 0.100 s with a one-rank "nccl" group in the process against 0.0947 without, and
# 0.0945 either way with sixteen (24: 0.0955; profiles/r06/rccl_queues.txt).  Every rank of a multi-GPU run has one.
def _request_hw_queues():
    """Sets GPU_MAX_HW_QUEUES = 16 unless the user has, and records whether the runtime can still see it: the variable
    is read once, by the first HIP call of the process.  If torch had already initialised the GPU when this package
    was imported, the setting is a silent no-op -- hw_queues() then says so and require_gpu() warns once."""
    import sys as _sys
    user = _os.environ.get("GPU_MAX_HW_QUEUES")
    late = False
    _torch = _sys.modules.get("torch")
    if _torch is not None:
        try:
            late = bool(_torch.cuda.is_initialized())
        except Exception:        # pragma: no cover
            late = False
    if user is None:
        _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    return {"value": int(user) if (user or "").isdigit() else (None if user else 16),
            "set_by": "user" if user is not None else "package",
            # None: unknown to this process (a user setting is the user's business); False: HIP was up before the import
            "in_effect": (None if user is not None else (not late))}


_HW_QUEUES = _request_hw_queues()


def hw_queues():
    """{'value', 'set_by': 'user' | 'package', 'in_effect'}: what GPU_MAX_HW_QUEUES this process asked for and whether
    the HIP runtime was still uninitialised when the package set it (False: the runtime's default of 4 applies and
    calc_probs on more than two streams runs ~10 % slower, profiles/r04/hw_queues.txt; import triceratops_amd before
    the first GPU call, or export the variable)."""
    return dict(_HW_QUEUES)


def set_sampling(mode):
    """'device' (default): the whole scenario on the GPU, the draw kernel's own Philox numbers (seeded through
    torch.manual_seed); 'numpy': priors sampled on the host from numpy's global stream, bit-for-bit the
    reference's host arithmetic under the same np.random.seed (the validation mode, ~300x slower at N = 1e6);
    'numpy-device': the same stream and draws, everything downstream of the uniforms on the GPU."""
    from .marginal_likelihoods import set_sampling as _set
    _set(mode)


def get_sampling():
    """the current sampling mode ('numpy', 'numpy-device' or 'device')"""
    from .marginal_likelihoods import _sampling
    return _sampling["mode"]


def set_precision(mode):
    """'fp64' (default) or 'fp32' (BASELINE config 5: fp32 flux model, fp64 orbit / chi^2 /
    log-mean-exp accumulators) for everything evaluated through lnZ_* and calc_probs."""
    from . import _lib
    _lib.set_precision(mode)


def set_full_evaluation(on):
    """True: every masked draw of every scenario is evaluated to the end (TRX_FLAG_FULL_EVALUATION) instead of the
    bounded evaluation; same results to rounding, several times slower on a real detection"""
    from . import _lib
    _lib.set_full_evaluation(bool(on))


def set_threads(n):
    """Host threads (each with its own HIP stream) that evaluate the scenarios of calc_probs /
    calc_probs_many side by side; effective with set_sampling("device") only (the numpy modes consume one
    global stream and stay on the calling thread).  With n > 1 every work unit draws from its own Philox key,
    so results do not depend on n >= 2 (sharding.per_unit_seed = True gives the same numbers with n = 1).
    One thread already overlaps the calls on sharding.streams HIP streams; more threads only help when
    the host side of the calls is the bottleneck."""
    from . import sharding
    if int(n) < 1:
        raise ValueError("threads must be >= 1")
    sharding.threads = int(n)


def calc_probs_many(jobs, verbose: int = 0):
    """calc_probs for several targets in one sharded pass (see triceratops.calc_probs_many)."""
    from .triceratops import calc_probs_many as _many
    return _many(jobs, verbose=verbose)
