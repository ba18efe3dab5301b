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
# profiles/r04_hw_queues.txt).  Read by the runtime when it initialises (the first HIP call of the process), so it
# has to be in the environment before that; a value the user has set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
