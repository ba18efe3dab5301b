"""The device path against the numbers the REFERENCE printed when it ran with the real pytransit.

No test of the reference evaluates a light curve, and pytransit 2.2 cannot run in this project (SURVEY.md
section 8c), so the stored cell outputs of examples/example.ipynb and examples/kepler_example.ipynb are the
only reference results that passed through pytransit's own transit arithmetic.  tests/golden/
notebook_anchors.npz (make_anchors.py) holds those tables next to the inputs of the cells that made them:
TOI-465.01 (100 binned points, with and without its contrast curve), TOI-411.02 (100 binned points) and
Kepler-10b (478 unbinned points, mission = "Kepler").  Each case runs here many times at N = 1e6 with
set_sampling("device") (calc_probs, 6-60 ms a run) and is held against the notebook:

 * the shares of TP : PTP : STP among themselves -- the three scenarios whose evidence needs no TRILEGAL
   population (a web query in the notebooks, a synthetic table here) -- in log space, within 3 sigma of this
   implementation's own seed-to-seed scatter (the notebook is ONE run with the same scatter);
 * the best-fit planet radius of the TP row, within 3 sigma of its scatter;
 * FPP over many runs against the notebook's "mean +- std of 20 runs" (cells 14 and 18), Welch's two-sample
   statistic.

What the notebooks cannot pin: they were made by an older release of the reference.  Its own tests document
fixes that postdate them (tests/test_background_prior_log_base.py: log10 -> ln in the background priors,
tests/test_beb_collision_mask.py), which move the D and B scenarios by construction, and TOI-465.01's FPP
without contrast curve comes out ten times lower with the CURRENT reference code as well: run in the build
container at N = 1e6 (profiles/reference_fpp_cpu.py, 6 runs, 95 s each; tests/golden/reference_runs.npz) it
gives FPP 0.001-0.011 where cell 14 printed 0.043 +- 0.058.  So the last test compares the device path with
those runs of the current code instead: same light curve, same star, same N, numpy's generator and the C
oracle on one side, Philox and the HIP kernels on the other.
Tables of a 300-seed run of every case: profiles/r03_notebook_anchors_300.txt (profiles/notebook_anchors.py)."""
import numpy as np
import pytest

import anchors
from helpers import gold

pytestmark = pytest.mark.gpu
N_RUNS = {"toi465_nocc": 64, "toi465_cc": 64, "toi411": 256, "kep10": 32}     # (toi411: the notebook sits 2.7-2.8 sigma off, see the 300-seed table: a tighter scatter estimate keeps the 3 sigma test off its own noise)
_cache = {}


def runs(case):
    if case not in _cache:
        _cache[case] = anchors.run_many(case, range(1000, 1000 + N_RUNS[case]))
    return _cache[case]


def welch(m1, s1, n1, m2, s2, n2):
    return (m1 - m2) / np.sqrt(s1 ** 2 / n1 + s2 ** 2 / n2)


@pytest.mark.parametrize("case", ["toi465_nocc", "toi411", "kep10"])
def test_trilegal_free_shares_against_the_notebook_table(case):
    lnZ, prob, fpp, rp = runs(case)
    nb_prob, _, _ = anchors.notebook(case)
    ours = np.log(anchors.free_shares(prob))
    want = np.log(anchors.free_shares(nb_prob)[0])
    sd = ours.std(axis=0, ddof=1) * np.sqrt(1 + 1 / ours.shape[0])
    z = (want - ours.mean(axis=0)) / sd
    print("\n%s TP:PTP:STP shares  ours %s  notebook %s  z %s" % (case, np.exp(ours.mean(axis=0)), np.exp(want), z))
    assert np.all(np.abs(z) < 3.0), (case, z)


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


@pytest.mark.xfail(strict=False, reason="cell 14 (0.0432 +- 0.0578) came from an older release: the current reference "
                   "code gives 0.006 +- 0.004 on the same input (reference_runs.npz), this implementation 0.0043 +- 0.0058")
def test_fpp_without_contrast_curve_against_the_notebooks_20_runs():
    fpp = runs("toi465_nocc")[2]
    m, s = anchors.A["toi465_FPP20_nocc"]
    t = welch(m, s, 20, fpp.mean(), fpp.std(ddof=1), fpp.size)
    print("\nTOI-465.01: FPP ours %.5f +- %.5f (%d runs), notebook %.4f +- %.4f (20 runs), Welch t %.2f"
          % (fpp.mean(), fpp.std(ddof=1), fpp.size, m, s, t))
    assert abs(t) < 3.0


@pytest.mark.parametrize("case", ["toi465_nocc", "toi411"])
def test_device_path_against_runs_of_the_current_reference_code(case):
    """lnZ of TP / PTP / STP, FPP and the TP radius: this implementation's runs against the reference's own
    (reference_runs.npz; 6 and 4 runs).  The evidences of a run are skewed (a lucky draw lifts lnZ), so the
    comparison is by rank: Mann-Whitney's two-sided p-value above 0.002 for each quantity."""
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
    print("\n%s vs %d runs of the reference code, Mann-Whitney p: %s; FPP ours %.5f (median %.5f), reference %.5f (median %.5f)"
          % (case, ref_fpp.size, {k: round(float(v), 3) for k, v in pv.items()}, fpp.mean(), np.median(fpp),
             ref_fpp.mean(), np.median(ref_fpp)))
    assert all(v > 0.002 for v in pv.values()), pv


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
