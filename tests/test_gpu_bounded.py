"""Bounded evaluation (cells_kernel<PRUNE>, include/trx.h: trx_set_bounded_evaluation).

trx_scenario_evidence keeps two things of the rows it evaluates: lnZ, the log-mean-exp of
c0 - chi^2/2 + lnprior (marginal_likelihoods.py:117-154), and the row with the smallest chi^2.  A row whose
chi^2 over the cells done so far shows that it can be neither is abandoned and reports that lower bound.
trx_set_debug_bounded_lnl(1) applies the same rule to trx_lnl_batch (as for an evidence without prior), which
makes the rule checkable row by row against the full evaluation."""
import ctypes

import numpy as np
import pytest
import torch

from triceratops_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _light_curve(n_time, rng, irregular=False):
    t = synth.time_grid(n_time)
    if irregular:
        t = np.sort(t + rng.uniform(-0.4, 0.4, n_time) * (t[1] - t[0]))
    t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    return t_d, _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()))


@pytest.mark.parametrize("n_time,irregular", [(100, False), (200, True), (500, True), (1500, False)])
def test_every_row_is_exact_or_a_valid_bound(n_time, irregular):
    """per row: the full chi^2/2, or a lower bound of it that lies above min + 90 (so that it carries no
    weight after the reduction's cut at 80 and cannot be the best draw); the minimum itself is always exact"""
    L = _lib.lib()
    rng = np.random.default_rng(11)
    t_d, f_d = _light_curve(n_time, rng, irregular)
    n = 30000 if n_time <= 500 else 6000
    cnt = ctypes.c_ulonglong(0)
    total = 0
    try:
        for fam in synth.FAMILIES:
            rows = synth.family_rows(rng, fam, n)
            rows[:, :3] = rows[:, 3:6]                  # ties
            rows[7 if fam[1] == _lib.MODEL_TP else 8, 5] = np.nan      # a draw with NaN eccentricity
            rows_d = _lib.dev(rows)
            flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
            L.trx_set_debug_bounded_lnl(0)
            full = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_set_debug_bounded_lnl(1)
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            got = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            total += cnt.value
            assert np.array_equal(np.isnan(full), np.isnan(got))
            assert np.array_equal(full == np.inf, got == np.inf)
            fin = np.isfinite(full)
            hmin = full[fin].min()
            exact = np.zeros(n, dtype=bool)
            exact[fin] = np.abs(got[fin] - full[fin]) <= 1e-11 * np.abs(full[fin])
            bound = fin & ~exact
            assert int(bound.sum()) == cnt.value, fam[0]
            assert np.all(got[bound] <= full[bound] * (1 + 1e-9))
            assert np.all(got[bound] > hmin + 90.0 - 1e-6)
            assert np.all(exact[fin & (full <= hmin + 90.0)])
            assert np.argmin(np.where(fin, got, np.inf)) == np.argmin(np.where(fin, full, np.inf))
    finally:
        L.trx_set_debug_bounded_lnl(0)
    assert total > 0.2 * n * len(synth.FAMILIES)        # the rule does bite on these families


@pytest.mark.parametrize("sigma", [2e-5, 5e-3])
def test_probe_pass_in_fp32_reports_valid_bounds_at_any_noise_level(sigma):
    """The probe pass of a split launch (batches of short light curves) evaluates its cells with the fp32 flux model
    whatever the call's precision (csrc/trx_kernels.hip: TRX_PROBE_FP32); what an abandoned row reports allows for the
    model's error, which weighs 1 / sigma in chi^2 (fp32_model_slack, csrc/trx_cells.hpp).  At 20 ppm noise -- where 1e-6
    of flux is a twentieth of a sigma per cell -- and at 5000 ppm: every reported value is the fp64 chi^2/2 or a lower
    bound of it above min + 90, and the rows that matter are exact."""
    L = _lib.lib()
    rng = np.random.default_rng(21)
    t = synth.time_grid(200)
    t_d = _lib.dev(t)
    curve, _ = _lib.flux_grid(0, 0, t_d, _lib.dev(synth.reference_tp_row()), synth.EXPTIME, 20, False)
    f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy(), sigma))
    n = 30000
    cnt = ctypes.c_ulonglong(0)
    total = 0
    try:
        for fam in synth.FAMILIES[:9]:
            rows_d = _lib.dev(synth.family_rows(rng, fam, n))
            flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
            L.trx_set_debug_bounded_lnl(0)
            full = _lib.lnl_batch(fam[1], flags, t_d, f_d, sigma, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_set_debug_bounded_lnl(1)
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            got = _lib.lnl_batch(fam[1], flags, t_d, f_d, sigma, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            total += cnt.value
            fin = np.isfinite(full)
            hmin = full[fin].min()
            exact = np.zeros(n, dtype=bool)
            exact[fin] = np.abs(got[fin] - full[fin]) <= 1e-11 * np.abs(full[fin])
            bound = fin & ~exact
            assert int(bound.sum()) == cnt.value, fam[0]
            assert np.all(got[bound] <= full[bound] * (1 + 1e-9)), fam[0]
            assert np.all(got[bound] > hmin + 90.0 - 1e-6), fam[0]
            assert np.all(exact[fin & (full <= hmin + 90.0)]), fam[0]
    finally:
        L.trx_set_debug_bounded_lnl(0)
    if sigma < 1e-3:
        assert total > 0.2 * n * 9          # (at 5000 ppm no row lies 90 above the best: nothing to abandon, every row exact)


def test_evidence_and_best_draw_do_not_depend_on_which_rows_stop():
    """lnZ and the best draw of trx_lnz_scenario with the bounded rows: equal to the full evaluation's to
    1e-13 (the probe cells of a row are summed first), and bit for bit the same from run to run although the
    set of abandoned rows changes with the timing of the waves"""
    L = _lib.lib()
    rng = np.random.default_rng(12)
    t_d, f_d = _light_curve(150, rng, True)
    n = 60000
    try:
        for fam in synth.FAMILIES[:8]:
            rows_d = _lib.dev(synth.family_rows(rng, fam, n))
            lp = _lib.dev(np.where(rng.random(n) < 0.1, -np.inf, -rng.exponential(3.0, n)))
            flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
            out = {}
            for mode in (0, 1, 1, 1):
                L.trx_set_debug_bounded_lnl(mode)
                h, lnz = _lib.lnz_scenario(fam[1], flags, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20, lp, n,
                                           float(np.log(synth.SIGMA)))
                out.setdefault(mode, []).append((float(lnz.cpu()[0]), int(torch.argmin(h).cpu())))
            (z0, b0), = out[0]
            assert all(b == b0 for _, b in out[1])
            assert all(abs(z - z0) <= 1e-13 * abs(z0) for z, _ in out[1])
    finally:
        L.trx_set_debug_bounded_lnl(0)


@pytest.mark.parametrize("n_time,rows_per_wave,packing_below", [(100, 22, None), (60, 16, None), (900, 0, 1000)])
def test_batches_that_span_several_window_passes_are_evaluated_in_full(n_time, rows_per_wave, packing_below):
    """The batched variant's verdict after the probe phase is only a lower bound when the batch's cells fit ONE
    768-cell window pass (its flat-model charge for the in-window cells not done yet covers the windows tested so
    far).  The public knobs can exceed that -- trx_set_rows_per_wave(22) at 100 points, a raised
    trx_set_cell_packing_below -- and such launches must not abandon anything: every row exact, none counted."""
    L = _lib.lib()
    rng = np.random.default_rng(13)
    t_d, f_d = _light_curve(n_time, rng, True)
    n = 20000
    cnt = ctypes.c_ulonglong(0)
    try:
        L.trx_set_rows_per_wave(rows_per_wave)
        if packing_below:
            L.trx_set_cell_packing_below(packing_below)
        for fam in synth.FAMILIES[:6]:
            rows_d = _lib.dev(synth.family_rows(rng, fam, n))
            flags = _lib.FLAG_COMPANION_IS_HOST if fam[2] else 0
            L.trx_set_debug_bounded_lnl(0)
            full = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_set_debug_bounded_lnl(1)
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            got = _lib.lnl_batch(fam[1], flags, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20).cpu().numpy()
            L.trx_pruned_rows(ctypes.byref(cnt), 1)
            assert cnt.value == 0, (fam[0], cnt.value)
            assert np.array_equal(full, got, equal_nan=True), fam[0]
    finally:
        L.trx_set_debug_bounded_lnl(0)
        L.trx_set_rows_per_wave(0)
        L.trx_set_cell_packing_below(_lib.CELL_PACKING_BELOW)


@pytest.mark.parametrize("chain", [0, 1])
@pytest.mark.parametrize("n_time,rows", [(200, 16), (200, 0), (150, 0), (150, 10), (100, 0), (64, 22)])
def test_probe_pass_with_its_own_rows_per_wave(n_time, rows, chain):
    """The probe pass of the batched bounded evaluation files only ~16 probe cells per row, so it takes more rows per
    wave than the passes that hold a batch's every cell (its own LDS layout; plan_cells, listed_pass_rows in
    csrc/trx_kernels.hip).  Which rows a WORKGROUP has (cells_entry's exit rule) and which rows its waves take
    (cells_body) must come from one rule: with the first version of this pass the two halved different starting values
    (16 -> 8 -> 4 -> 2 against 3 -> 2) and, call by call, landed on 2 and 3 rows per wave for ~12 000 listed rows -- the
    last batches had no workgroup, and the "never written" status of the record said so at once.  Several N so that
    the listed counts cross the steps of the rule; the records must be those of the pass at the other passes' rows per
    wave, bit for bit (results do not depend on how rows are grouped: the bounded instantiations freeze settled lanes)."""
    import triceratops_amd
    from triceratops_amd import _lib, sharding, synth
    from helpers import GOLD
    import os
    import torch
    L = _lib.lib()
    tri, cc = os.path.join(GOLD, "trilegal_synth.csv"), os.path.join(GOLD, "contrast_curve_synth.csv")
    saved = sharding.streams
    triceratops_amd.set_sampling("device")
    try:
        L.trx_set_star_chain(chain)
        sharding.streams = 2
        for N in (120_000, 260_000, 1_000_000):
            out = []
            for r in (1, rows):
                assert L.trx_set_probe_rows(r) == 0
                jobs = synth.toi_jobs(2, n_time=n_time, N=N, seed=3, trilegal_fname=tri, contrast_curve_file=cc)
                torch.manual_seed(5)
                triceratops_amd.calc_probs_many(jobs)          # (TrxError if a row was never written)
                out.append([np.concatenate([tg.lnZ, [tg.FPP, tg.NFPP]]) for tg, _ in jobs])
            for x, y in zip(*out):
                assert np.array_equal(x, y, equal_nan=True), (n_time, rows, N)
    finally:
        L.trx_set_probe_rows(0)
        L.trx_set_star_chain(1)
        sharding.streams = saved
        triceratops_amd.set_sampling("numpy")
