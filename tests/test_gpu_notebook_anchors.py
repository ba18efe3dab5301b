"""The device path against the numbers the REFERENCE printed when it ran with the real pytransit.

No test of the reference evaluates a light curve, and pytransit 2.2 cannot run in this project (SURVEY.md
section 8c), so the stored cell outputs of examples/example.ipynb and examples/kepler_example.ipynb are the
only reference results that passed through pytransit's own transit arithmetic.  tests/golden/
notebook_anchors.npz (make_anchors.py) holds those tables next to the inputs of the cells that made them:
TOI-465.01 (100 binned points, with and without its contrast curve), TOI-411.02 (100 binned points) and
Kepler-10b (478 unbinned points, mission = "Kepler").  Each case runs here N_RUNS = 64 times at N = 1e6 with
set_sampling("device") (calc_probs, 6-60 ms a run; the count is the same for every case and was fixed before
the runs were looked at) and is held against

 * the CURRENT reference code: tests/golden/reference_runs.npz, the reference's own calc_probs run on the CPU of
   the build container at N = 1e6 with the oracle at pytransit's seam (profiles/reference_fpp_cpu.py, ~90 s a
   run: 16 runs of each of TOI-465.01 and TOI-411.02).  This is where agreement is REQUIRED: lnZ of TP / PTP / STP, FPP
   and the TP radius by rank (Mann-Whitney), the log-shares ln(PTP/TP), ln(STP/TP) of TOI-411.02 -- the one
   tight anchor, 0.07 and 0.11 of scatter per run -- within 3 standard errors of the difference of the means;
 * the notebook's single run: the shares of TP : PTP : STP among themselves (the three scenarios whose evidence
   needs no TRILEGAL population) in log space, and the best-fit planet radius of the TP row, within 3 sigma of a
   run's scatter -- for TOI-465.01 and Kepler-10b, whose scatter makes that a sanity check (0.6-1.4 and 2.5-8);
 * FPP over many runs against the notebook's "mean +- std of 20 runs" with the contrast curve (cell 18), Welch's
   statistic.

What the notebooks do NOT pin, and is therefore reported, not gated:
 * TOI-411.02's TP : PTP : STP (cell 25: 0.751 : 0.119 : 0.0338; here 0.797 : 0.156 : 0.0448, the current reference
   code 0.80 : 0.15 : 0.046).  In log space the notebook sits -0.214 (PTP/TP) and -0.221 (STP/TP) from this
   implementation -- the SAME factor 0.81 on both bound-companion scenarios.  profiles/r04/anchor_sensitivity.txt
   changes every input this repository had to make up (lightkurve's bin edges, the sigma it reports, the last
   printed digit of the star table, sampling mode, parallel, exposure settings, N) one at a time over 100 paired
   seeds: none moves ln(PTP/TP) by more than 0.02.  A transit-arithmetic difference would not hit PTP and STP
   alike (the planet orbits different stars); a constant of the bound-companion prior does.  Tested in round 5
   (profiles/r05/anchor_prior_forms.txt): the earlier form of that prior, which priors.py:661-688, 749-776 keeps as
   comments, moves both ratios by +0.481 -- the WRONG direction; a separation limit of 1.1 arcsec instead of the
   2.2 arcsec that stands in for a contrast curve moves both by -0.2188 -- the notebook's offset to 0.005, and not
   checkable against any older release here.  Status: unexplained.  The offsets are REPORTED; what is asserted about
   them is only "no gross error" (each below 0.5) and that they are one common factor.
 * TOI-465.01's FPP without contrast curve (cell 14: 0.0432 +- 0.0578 over 20 runs): the CURRENT reference code
   gives 0.001-0.011 on the same input (reference_runs.npz), this implementation 0.005 +- 0.006 -- printed next
   to the rank test against those runs.  The reference's own tests document fixes that postdate the notebooks
   (tests/test_background_prior_log_base.py: log10 -> ln in the background priors; tests/test_beb_collision_mask.py),
   which move the D and B scenarios by construction.
Tables of a 300-seed run of every case: profiles/r03/notebook_anchors_300.txt (profiles/notebook_anchors.py)."""
import numpy as np
import pytest

import anchors
from helpers import gold

pytestmark = pytest.mark.gpu
N_RUNS = 64           # every case; fixed before looking at a run
SEEDS = range(1000, 1000 + N_RUNS)


def runs(case):
    return anchors.run_many(case, SEEDS)


def welch(m1, s1, n1, m2, s2, n2):
    return (m1 - m2) / np.sqrt(s1 ** 2 / n1 + s2 ** 2 / n2)


@pytest.mark.parametrize("case", ["toi465_nocc", "kep10"])
def test_trilegal_free_shares_against_the_notebook_table(case):
    lnZ, prob, fpp, rp = runs(case)
    nb_prob, _, _ = anchors.notebook(case)
    ours = np.log(anchors.free_shares(prob))
    want = np.log(anchors.free_shares(nb_prob)[0])
    sd = ours.std(axis=0, ddof=1) * np.sqrt(1 + 1 / ours.shape[0])
    z = (want - ours.mean(axis=0)) / sd
    print("\n%s TP:PTP:STP shares  ours %s  notebook %s  z %s" % (case, np.exp(ours.mean(axis=0)), np.exp(want), z))
    assert np.all(np.abs(z) < 3.0), (case, z)


def _log_ratios(lnZ3):
    """ln(PTP/TP), ln(STP/TP) from [runs][TP, PTP, STP] evidences"""
    return np.stack([lnZ3[:, 1] - lnZ3[:, 0], lnZ3[:, 2] - lnZ3[:, 0]], axis=1)


def test_toi411_log_shares_equal_the_current_reference_code_and_the_notebook_offset_is_one_common_factor():
    """TOI-411.02, the one tight anchor (module docstring).  Required: ln(PTP/TP) and ln(STP/TP) of this
    implementation = those of the reference's current code (16 CPU runs) within 3 standard errors.  Reported and
    checked for its structure only: the notebook's single run sits the same distance below on both."""
    R = gold("reference_runs.npz")
    lnZ, prob, fpp, rp = runs("toi411")
    cols = [anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP")]
    ours, ref = _log_ratios(lnZ[:, cols]), _log_ratios(R["toi411_lnZ"])
    se = np.sqrt(ours.var(axis=0, ddof=1) / ours.shape[0] + ref.var(axis=0, ddof=1) / ref.shape[0])
    z = (ours.mean(axis=0) - ref.mean(axis=0)) / se
    nb_prob, _, _ = anchors.notebook("toi411")
    nb = np.log(np.array([nb_prob[cols[1]] / nb_prob[cols[0]], nb_prob[cols[2]] / nb_prob[cols[0]]]))
    off = nb - ours.mean(axis=0)
    sd = ours.std(axis=0, ddof=1)
    print("\ntoi411 ln(PTP/TP), ln(STP/TP): ours %s +- %s (a run), reference code %s (%d runs), z %s; notebook %s, "
          "offset %s = %s sigma of a run" % (ours.mean(axis=0), sd, ref.mean(axis=0), ref.shape[0], z, nb, off, off / sd))
    assert np.all(np.abs(z) < 3.0), z
    # the notebook: one common factor on both bound-companion scenarios (not gated on its size beyond "no gross error")
    assert np.all(np.abs(off) < 0.5), off
    assert abs(off[0] - off[1]) < 3.0 * np.sqrt(sd[0] ** 2 + sd[1] ** 2), off


@pytest.mark.parametrize("case", ["toi465_nocc", "toi411", "kep10"])
def test_best_fit_planet_radius_against_the_notebook(case):
    lnZ, prob, fpp, rp = runs(case)
    _, _, nb_rp = anchors.notebook(case)
    z = (nb_rp - rp.mean()) / (rp.std(ddof=1) * np.sqrt(1 + 1 / rp.size))
    print("\n%s best-fit TP R_p: ours %.3f +- %.3f, notebook %.3f (z %.2f)" % (case, rp.mean(), rp.std(ddof=1), nb_rp, z))
    assert abs(z) < 3.0


def test_fpp_with_contrast_curve_against_the_notebooks_20_runs():
    """examples/example.ipynb cell 18: FPP = 0.0032 +- 0.005 over 20 runs"""
    fpp = runs("toi465_cc")[2]
    m, s = anchors.A["toi465_FPP20_cc"]
    t = welch(m, s, 20, fpp.mean(), fpp.std(ddof=1), fpp.size)
    print("\nTOI-465.01 + contrast curve: FPP ours %.5f +- %.5f (%d runs), notebook %.4f +- %.4f (20 runs), Welch t %.2f"
          % (fpp.mean(), fpp.std(ddof=1), fpp.size, m, s, t))
    assert abs(t) < 3.0


@pytest.mark.parametrize("case", ["toi465_nocc", "toi411"])
def test_device_path_against_runs_of_the_current_reference_code(case):
    """lnZ of TP / PTP / STP, FPP and the TP radius: this implementation's runs against the reference's own
    (reference_runs.npz; 16 runs each).  The evidences of a run are skewed (a lucky draw lifts lnZ), so the
    comparison is by rank: Mann-Whitney's two-sided p-value above 0.002 for each quantity.  The notebook's FPP
    of the same input is printed beside it (cell 14 / cell 25): reported, not gated (module docstring)."""
    from scipy.stats import mannwhitneyu
    R = gold("reference_runs.npz")
    lnZ, prob, fpp, rp = runs(case)
    ref_lnZ, ref_fpp, ref_rp = R[case + "_lnZ"], R[case + "_FPP"], R[case + "_Rp"]
    cols = [anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP")]
    pv = {}
    for j, name in zip(range(3), ("lnZ TP", "lnZ PTP", "lnZ STP")):
        pv[name] = mannwhitneyu(lnZ[:, cols[j]], ref_lnZ[:, j], alternative="two-sided").pvalue
    pv["FPP"] = mannwhitneyu(fpp, ref_fpp, alternative="two-sided").pvalue
    pv["R_p"] = mannwhitneyu(rp, ref_rp, alternative="two-sided").pvalue
    nb = ("notebook cell 14, 20 runs of an older release: %.4f +- %.4f" % tuple(anchors.A["toi465_FPP20_nocc"])
          if case == "toi465_nocc" else "notebook cell 25, one run: %.4f" % anchors.notebook(case)[1])
    print("\n%s vs %d runs of the reference code, Mann-Whitney p: %s; FPP ours %.5f +- %.5f (median %.5f), reference "
          "code %.5f +- %.5f (median %.5f); %s"
          % (case, ref_fpp.size, {k: round(float(v), 3) for k, v in pv.items()}, fpp.mean(), fpp.std(ddof=1),
             np.median(fpp), ref_fpp.mean(), ref_fpp.std(ddof=1), np.median(ref_fpp), nb))
    assert all(v > 0.002 for v in pv.values()), pv
    # the FPP of the current reference code is the one this implementation must reproduce (Welch on the means)
    t = welch(fpp.mean(), fpp.std(ddof=1), fpp.size, ref_fpp.mean(), ref_fpp.std(ddof=1), ref_fpp.size)
    assert abs(t) < 3.0, t


def test_numpy_mode_replays_a_run_of_the_reference_code_at_full_size():
    """set_sampling("numpy") consumes numpy's global stream exactly as the reference does, so seed 1000 of
    reference_runs.npz (TOI-465.01 without contrast curve, N = 1e6, the reference's calc_probs on the CPU with
    the oracle behind pytransit's seam) must come out again on the GPU: lnZ to the 1e-3 the run was logged
    with, FPP and the TP radius to the digits logged."""
    R = gold("reference_runs.npz")
    assert int(R["toi465_nocc_seed"][0]) == 1000 and int(R["toi465_nocc_N"][0]) == 1_000_000
    lnZ, prob, fpp, rp = anchors.run("toi465_nocc", 1000, sampling="numpy")
    cols = [anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP")]
    print("\nnumpy mode, seed 1000: lnZ %s FPP %.5f R_p %.3f; reference run: lnZ %s FPP %.5f R_p %.3f"
          % (lnZ[cols], fpp, rp, R["toi465_nocc_lnZ"][0], R["toi465_nocc_FPP"][0], R["toi465_nocc_Rp"][0]))
    assert np.allclose(lnZ[cols], R["toi465_nocc_lnZ"][0], rtol=0, atol=2e-3)
    assert abs(fpp - R["toi465_nocc_FPP"][0]) < 2e-5 and abs(rp - R["toi465_nocc_Rp"][0]) < 2e-3
