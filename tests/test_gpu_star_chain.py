"""One launch chain per group of lnZ_* calls (trx_star_enqueue, include/trx.h; csrc/trx_scenario.hip enqueue_chain).

The calls of a target that share N and the time stamps run as ONE chain of launches -- every kernel of the path once,
the call (draw side) or the branch (likelihood side) as a further grid dimension -- instead of 9-17 launches per call.
The kernels' bodies are the per-call chain's, so the records must be those of the per-call chain BIT FOR BIT:
 * synthetic targets (target + one nearby star: 12 calls, 18 branches, one chain each) at 100, 200 (batches of rows per
   wave) and 478 points (one row per wave), device-side random numbers;
 * TOI-465.01's 75-scenario blend (50 calls in five chains) on one and four streams;
 * the reference's seeded draws through the chain (numpy-device; the fixtures themselves: test_gpu_production_pin.py);
and the record's "never written" status must catch round 4's exit-rule bug when it is switched back on
(trx_set_debug_bug), through the chain and call by call.
Reference: triceratops.py:767-1428 (the scenario loop of calc_probs)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import GOLD

pytestmark = pytest.mark.gpu
TRI = os.path.join(GOLD, "trilegal_synth.csv")
CC = os.path.join(GOLD, "contrast_curve_synth.csv")


def _tables(jobs):
    out = []
    for tg, _ in jobs:
        out.append(np.concatenate([tg.lnZ, tg.probs["prob"].values, tg.probs["R_p"].values, tg.probs["inc"].values,
                                   tg.probs["M_EB"].values, tg.u1, tg.fluxratio_comp, [tg.FPP, tg.NFPP]]))
    return out


def _many(n_tois, n_time, N, seed, chain, streams=None):
    import triceratops_amd
    from triceratops_amd import _lib, sharding, synth
    L = _lib.lib()
    saved = sharding.streams
    triceratops_amd.set_sampling("device")
    try:
        L.trx_set_star_chain(1 if chain else 0)
        if streams:
            sharding.streams = streams
        jobs = synth.toi_jobs(n_tois, n_time=n_time, N=N, seed=seed, trilegal_fname=TRI, contrast_curve_file=CC)
        torch.manual_seed(seed)
        _lib.reset_stats()
        triceratops_amd.calc_probs_many(jobs)
        assert _lib.STATS["native_calls"] == 12 * n_tois
        return _tables(jobs)
    finally:
        L.trx_set_star_chain(1)
        sharding.streams = saved
        triceratops_amd.set_sampling("numpy")


@pytest.mark.parametrize("n_time,N", [(100, 200_000), (200, 1_000_000), (478, 300_000), (60, 50_000)])
def test_chain_records_equal_the_per_call_chain_bit_for_bit(n_time, N):
    a = _many(3, n_time, N, 11, chain=True)
    b = _many(3, n_time, N, 11, chain=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True), np.nanmax(np.abs(x - y))
    assert np.isfinite(a[0][0]) and -1e-12 <= a[0][-2] <= 1.0


def test_chain_on_one_and_on_four_streams():
    a = _many(5, 200, 300_000, 5, chain=True, streams=1)
    b = _many(5, 200, 300_000, 5, chain=True, streams=4)
    c = _many(5, 200, 300_000, 5, chain=False, streams=3)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True)


def test_short_light_curves_fall_back_to_the_per_call_chain():
    """fewer than 48 points: no bounded evaluation, hence no chain -- the calls go one by one and still agree"""
    a = _many(2, 40, 50_000, 3, chain=True)
    b = _many(2, 40, 50_000, 3, chain=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def _blend(N, seed, sampling="device"):
    import test_toi465 as T
    import triceratops_amd
    triceratops_amd.set_sampling(sampling)
    try:
        torch.manual_seed(seed)
        return T._run("blend", N, seed)
    finally:
        triceratops_amd.set_sampling("numpy")


def test_blend_of_75_scenarios_chain_equals_per_call():
    from triceratops_amd import _lib, sharding
    L = _lib.lib()
    saved = sharding.streams
    try:
        runs = {}
        for chain, streams in ((0, 1), (1, 1), (1, 4), (1, 6)):
            L.trx_set_star_chain(chain)
            sharding.streams = streams
            runs[(chain, streams)] = _blend(1_000_000, 465)
        ref = runs[(0, 1)]
        assert np.isfinite(ref.lnZ).sum() >= 60
        for key, tg in runs.items():
            assert np.array_equal(tg.lnZ, ref.lnZ, equal_nan=True), (key, np.nanmax(np.abs(tg.lnZ - ref.lnZ)))
            assert tg.FPP == ref.FPP and tg.NFPP == ref.NFPP
            assert np.array_equal(tg.probs["R_p"].values, ref.probs["R_p"].values)
    finally:
        L.trx_set_star_chain(1)
        sharding.streams = saved


def test_seeded_reference_draws_through_the_chain():
    """numpy-device: the reference's uniforms, staged, through the chain = through the calls one by one"""
    from triceratops_amd import _lib
    L = _lib.lib()
    try:
        L.trx_set_star_chain(1)
        a = _blend(200_000, 77, "numpy-device")
        L.trx_set_star_chain(0)
        b = _blend(200_000, 77, "numpy-device")
    finally:
        L.trx_set_star_chain(1)
    assert np.array_equal(a.lnZ, b.lnZ, equal_nan=True) and a.FPP == b.FPP


@pytest.mark.parametrize("chain", [1, 0])
def test_rows_never_written_are_reported_not_read(chain):
    """Round 4's first three-pass scheme took the rows per wave of the third pass's workgroup EXIT rule from the whole
    row count and those of the pass itself from the rows behind the pilot: when nothing was probed and the
    two counts fell on different sides of a step of batch_rows, the last batches were never written, the stale chi^2 of
    the stream's previous call read as a result, and the 75-scenario blend's FPP came out as 1.  trx_set_debug_bug(1)
    switches that exit rule back on.  TOI-411.02 (a 166 ppm signal: the pilot's verdict is "probing does not pay") at
    N = 3e5 has scenarios on such a step call by call (profiles/r05/scan_debug_bug.txt: 18 of 27 values of N between
    1.5e5 and 4.1e5 do; the rule asks for 3200 waves a launch), and at N = 3e4 in a launch chain (every N from 1e4 to 9e4:
    a chain's rule asks for 3200 / branches waves a branch).  The run must now FAIL with TrxError -- record status 1, a
    row still carries rowc_kernel's mark -- instead of returning numbers, through the chain and call by call."""
    N = 30_000 if chain else 300_000
    import anchors
    from triceratops_amd import _lib, sharding
    L = _lib.lib()
    saved = sharding.streams
    try:
        L.trx_set_star_chain(chain)
        sharding.streams = 4
        good = anchors.run("toi411", 7, N=N, sampling="device")
        L.trx_set_debug_bug(1)
        with pytest.raises(_lib.TrxError, match="no kernel wrote"):
            anchors.run("toi411", 7, N=N, sampling="device")
        L.trx_set_debug_bug(0)
        again = anchors.run("toi411", 7, N=N, sampling="device")
        assert np.array_equal(good[0], again[0], equal_nan=True) and good[2] == again[2]
    finally:
        L.trx_set_debug_bug(0)
        L.trx_set_star_chain(1)
        sharding.streams = saved
