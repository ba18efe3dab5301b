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
N_RUNS = {"toi465_nocc": 64, "toi465_cc": 64, "toi411": 64, "kep10": 32}
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
    """lnZ of TP / PTP / STP, log FPP and the TP radius: this implementation's runs against the reference's own
    (reference_runs.npz), Welch's statistic below 3 for each"""
    R = gold("reference_runs.npz")
    lnZ, prob, fpp, rp = runs(case)
    ref_lnZ, ref_fpp, ref_rp = R[case + "_lnZ"], R[case + "_FPP"], R[case + "_Rp"]
    cols = [anchors.SCENARIOS.index(s) for s in ("TP", "PTP", "STP")]
    stats = {}
    for j, name in zip(range(3), ("lnZ TP", "lnZ PTP", "lnZ STP")):
        a, b = lnZ[:, cols[j]], ref_lnZ[:, j]
        stats[name] = welch(a.mean(), a.std(ddof=1), a.size, b.mean(), b.std(ddof=1), b.size)
    a, b = np.log(fpp), np.log(ref_fpp)
    stats["ln FPP"] = welch(a.mean(), a.std(ddof=1), a.size, b.mean(), b.std(ddof=1), b.size)
    stats["R_p"] = welch(rp.mean(), rp.std(ddof=1), rp.size, ref_rp.mean(), ref_rp.std(ddof=1), ref_rp.size)
    print("\n%s vs %d runs of the reference code: %s; FPP ours %.5f (median %.5f), reference %.5f"
          % (case, ref_fpp.size, {k: round(float(v), 2) for k, v in stats.items()}, fpp.mean(), np.median(fpp), ref_fpp.mean()))
    assert all(abs(v) < 3.0 for v in stats.values()), stats
