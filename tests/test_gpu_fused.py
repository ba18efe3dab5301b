"""The fused per-draw kernel (trx_draw_scenario, triceratops_amd/fused.py) against the elementwise
torch expression of the same chain (tests/torch_pipeline.py, itself pinned to the reference's goldens):
the same staged random numbers through both must give the same masks, columns, priors and
therefore lnZ and best-fit tables -- for all ten scenarios, with and without a contrast curve,
fixed period and period range, vector-path and per-draw-loop semantics, MOLUSC table."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import GOLD, gold

pytestmark = pytest.mark.gpu
G = gold("lnz_cases.npz")
TRI = os.path.join(GOLD, "trilegal_synth.csv")
CC = os.path.join(GOLD, "contrast_curve_synth.csv")
MOL = os.path.join(GOLD, "molusc_synth.csv")
NAMES = ("TTP", "TEB", "PTP", "PEB", "STP", "SEB", "DTP", "DEB", "BTP", "BEB")


def _call(ml, name, P, N, parallel, cc, filt, star=(0.82, 0.8, 5100.0), molusc=None):
    M_s, R_s, Teff = star
    base = (G["time"], G["flux"], float(G["sigma"][0]), P, M_s, R_s, Teff)
    fn = getattr(ml, "lnZ_" + name)
    if name in ("TTP", "TEB"):
        return fn(*base, 0.0, N, parallel)
    if name in ("PTP", "PEB", "STP", "SEB"):
        return fn(*base, 0.0, 14.2, cc, filt, N, parallel, "TESS", False, 0.00139, 20, molusc)
    mags = (10.4, 9.5, 9.1, 9.0)
    if name in ("DTP", "DEB"):
        return fn(*base, 0.0, *mags, TRI, cc, filt, N, parallel)
    return fn(*base, *mags, TRI, cc, filt, N, parallel)


def _both(name, mode, seed, **kw):
    """the product's fused kernel (through the lnZ_* dispatch) and the torch expression of the same chain
    (tests/torch_pipeline.py) on the same random-number source"""
    import torch_pipeline
    import triceratops_amd
    from triceratops_amd import marginal_likelihoods as ml
    out = {}
    triceratops_amd.set_sampling(mode)
    try:
        for fused in (True, False):
            np.random.seed(seed)
            torch.manual_seed(seed)
            out[fused] = _call(ml if fused else torch_pipeline, name, **kw)
    finally:
        triceratops_amd.set_sampling("numpy")
    return out[True], out[False]


def _same(a, b, lnz_tol, n_rows):
    da = a if isinstance(a, tuple) else (a,)
    db = b if isinstance(b, tuple) else (b,)
    assert len(da) == len(db)
    for x, y in zip(da, db):
        assert (x["lnZ"] == y["lnZ"]) if not np.isfinite(y["lnZ"]) else abs(x["lnZ"] - y["lnZ"]) < lnz_tol + 1e-13 * abs(y["lnZ"])
        for k in y:
            if k == "lnZ":
                continue
            assert np.allclose(np.asarray(x[k])[:n_rows], np.asarray(y[k])[:n_rows], rtol=1e-9, atol=1e-300,
                               equal_nan=True), k


@pytest.mark.parametrize("variant", ["fixed", "range", "ccJ", "ccTESS", "serial", "flat"])
@pytest.mark.parametrize("name", NAMES)
def test_fused_kernel_equals_torch_pipeline_on_the_numpy_stream(name, variant):
    kw = dict(P=3.3, N=40000, parallel=True, cc=None, filt="TESS")
    if variant == "range":
        kw["P"] = [2.5, 4.0]
    elif variant == "ccJ":
        kw.update(cc=CC, filt="J")
    elif variant == "ccTESS":
        kw.update(cc=CC, filt="TESS")
    elif variant == "serial":
        kw.update(parallel=False, N=5000)
    elif variant == "flat":
        kw.update(star=(0.4, 0.42, 3500.0))          # M <= 0.45: the other planet-radius law, q_min = 0.25
    a, b = _both(name, "numpy-device", 31 + NAMES.index(name), **kw)
    _same(a, b, 1e-9, 100)


@pytest.mark.parametrize("name", ["PTP", "PEB", "STP", "SEB"])
def test_fused_kernel_with_a_molusc_table(name):
    a, b = _both(name, "numpy-device", 5, P=3.3, N=900, parallel=True, cc=None, filt="TESS", molusc=MOL)
    _same(a, b, 1e-9, 50)


@pytest.mark.parametrize("name", NAMES)
def test_fused_kernel_equals_torch_pipeline_on_the_device_generator(name):
    """torch's device generator staged for both paths and consumed in the same order (fused.PHILOX
    off: with it on the kernel draws its own numbers, tested further down)"""
    from triceratops_amd import fused
    fused.PHILOX = False
    try:
        a, b = _both(name, "device", 77, P=3.3, N=200000, parallel=True, cc=CC, filt="J")
    finally:
        fused.PHILOX = True
    _same(a, b, 1e-9, 100)


def test_missing_limb_darkening_cell_raises_like_the_reference():
    """SEB draws companions up to 13000 K but the Claret grid stops at 10000 K: a hot target makes
    the reference's `.item()` on the empty match raise ValueError; so do both device paths"""
    import torch_pipeline
    import triceratops_amd
    from triceratops_amd import marginal_likelihoods as ml
    triceratops_amd.set_sampling("device")
    try:
        for mod in (ml, torch_pipeline):
            torch.manual_seed(3)
            with pytest.raises(ValueError):
                _call(mod, "SEB", 3.3, 20000, True, None, "TESS", star=(4.0, 2.9, 14000.0))
    finally:
        triceratops_amd.set_sampling("numpy")


# ---------------------------------------------------------------------------------------------
# the kernel's own random numbers (Philox4x32-10, set_sampling("device"))
class _Replay:
    """hands the draw kernel's dumped random numbers back, as staged arrays, in the order a
    scenario consumes them (device_pipeline.TorchRng interface)"""

    def __init__(self, dump, order):
        self.dump, self.order = dump, list(order)

    def uniform(self, n, device):
        return self.dump[self.order.pop(0)].clone()

    def beta(self, n, a, b, device):
        return self.dump[8].clone()

    def randint(self, hi, n, device):
        return self.dump[7].to(torch.int64)

    def discard(self, n):
        pass


_ORDER = {"TTP": [2, 3, 6], "PTP": [1, 2, 3, 6], "STP": [1, 2, 3, 6], "DTP": [2, 3, 6], "BTP": [2, 3, 6],
          "TEB": [3, 4, 5, 6], "PEB": [3, 4, 5, 6, 1], "SEB": [3, 4, 5, 6, 1], "DEB": [3, 4, 5, 6], "BEB": [3, 4, 5, 6]}


@pytest.mark.parametrize("P", [3.3, [2.5, 4.0]], ids=["fixed", "range"])
@pytest.mark.parametrize("name", NAMES)
def test_in_kernel_random_numbers_equal_the_staged_path_on_the_same_numbers(name, P):
    """device mode: the kernel draws its own numbers; dumped and fed back through the staged path
    (the one pinned to the reference) they give the same evidence and tables"""
    import triceratops_amd
    from triceratops_amd import device_pipeline as dp
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    kw = dict(P=P, N=60000, parallel=True, cc=CC, filt="J")
    triceratops_amd.set_sampling("device")
    fused.DUMP = []
    try:
        torch.manual_seed(123)
        a = _call(ml, name, **kw)
        rec = fused.DUMP[0]
        order = ([0] if isinstance(P, list) else []) + _ORDER[name]
        fused.DUMP = None
        fused.PHILOX = False
        saved = dp.RNG
        dp.RNG = _Replay(rec["dump"], order)
        try:
            b = _call(ml, name, **kw)
        finally:
            dp.RNG = saved
    finally:
        fused.DUMP, fused.PHILOX = None, True
        triceratops_amd.set_sampling("numpy")
    if isinstance(P, list):
        # the staged path hands sample_ecc the SAMPLE mean of the periods, the kernel path the expectation:
        # the same side of the 10-day switch of the binaries' eccentricity law in this test
        pass
    _same(a, b, 1e-10, 100)


def test_in_kernel_random_numbers_statistics():
    """uniformity, independence between slots and draws, the index draw and the Beta(0.867, 3.030)
    eccentricities (inverse CDF of the slot-5 uniform, csrc/trx_draw.hip: ecc_from_uniform) against scipy"""
    from scipy import stats
    import triceratops_amd
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    N = 400000
    triceratops_amd.set_sampling("device")
    fused.DUMP = []
    try:
        torch.manual_seed(9)
        _call(ml, "BTP", P=[2.5, 4.0], N=N, parallel=True, cc=None, filt="TESS")     # P, index, R_p, inc, beta, argp
        _call(ml, "PEB", P=3.3, N=N, parallel=True, cc=None, filt="TESS")            # inc, q, ecc, argp, q_c
        d1, d2 = (r["dump"].cpu().numpy() for r in fused.DUMP)
    finally:
        fused.DUMP = None
        triceratops_amd.set_sampling("numpy")
    streams = {"P": d1[0], "Rp": d1[2], "inc": d1[3], "w": d1[6], "inc2": d2[3], "q": d2[4], "ecc": d2[5],
               "w2": d2[6], "qc": d2[1]}
    for k, u in streams.items():
        assert u.min() >= 0.0 and u.max() < 1.0, k
        assert abs(u.mean() - 0.5) < 5 * np.sqrt(1 / 12 / N) and abs(u.var() - 1 / 12) < 5e-4, k
        assert stats.kstest(u, "uniform").pvalue > 1e-4, k
        assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(N), k          # draw i vs draw i + 1
    keys = list(streams)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            assert abs(np.corrcoef(streams[keys[i]], streams[keys[j]])[0, 1]) < 5 / np.sqrt(N), (keys[i], keys[j])
    idx = d1[7]
    n_field = int(idx.max()) + 1
    counts = np.bincount(idx.astype(int), minlength=n_field)
    assert stats.chisquare(counts).pvalue > 1e-4 and n_field > 100
    ecc = d1[8]
    assert stats.kstest(ecc, stats.beta(0.867, 3.030).cdf).pvalue > 1e-4
    # ... and draw by draw: the exact quantile of the uniform the draw used, to the table's 1e-6
    assert np.abs(ecc - stats.beta(0.867, 3.030).ppf(d1[5])).max() < 2e-6
    assert stats.kstest(d1[5], "uniform").pvalue > 1e-4 and abs(np.corrcoef(d1[5], d1[6])[0, 1]) < 5 / np.sqrt(N)
    assert abs(ecc.mean() - 0.867 / (0.867 + 3.030)) < 5 * stats.beta(0.867, 3.030).std() / np.sqrt(N)


def test_in_kernel_random_numbers_depend_on_seed_and_draw_index_only():
    import triceratops_amd
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    triceratops_amd.set_sampling("device")
    out = []
    try:
        for seed, N in ((5, 3000), (5, 7000), (6, 3000)):
            fused.DUMP = []
            torch.manual_seed(seed)
            _call(ml, "TTP", P=3.3, N=N, parallel=True, cc=None, filt="TESS")
            out.append(fused.DUMP[0]["dump"].cpu().numpy())
    finally:
        fused.DUMP = None
        triceratops_amd.set_sampling("numpy")
    assert np.array_equal(out[0], out[1][:, :3000])          # same seed: the first draws do not depend on N
    assert not np.array_equal(out[0][2], out[2][2])           # another seed: other numbers


@pytest.mark.parametrize("star", [(0.82, 0.8, 5100.0), (0.44, 0.43, 3600.0)], ids=["K", "M"])
@pytest.mark.parametrize("name", NAMES)
def test_fp32_pretest_of_the_geometry_leaves_the_masks_unchanged(name, star):
    """trx_scenario_enqueue evaluates the fp64 geometry mask only for the draws that pass may_transit(), an fp32
    necessary condition (csrc/trx_draw.hip).  With and without it: the same number of masked draws in either
    branch, the same lnZ bits, the same best draw -- at N = 1e6, fixed period and period range, both `parallel`
    semantics, a K dwarf and an M dwarf below the 0.45 M_sun switch of the planet-radius law."""
    import triceratops_amd
    from triceratops_amd import _lib, fused
    from triceratops_amd import marginal_likelihoods as ml
    triceratops_amd.set_sampling("device")
    saved_rows, saved_pre = fused.TABLE_ROWS, fused.PRETEST
    fused.TABLE_ROWS = 1                       # the native scenario call (what calc_probs uses)
    try:
        for P in (3.3, [2.5, 4.0]):
            for parallel in (True, False):
                out = []
                for pre in (True, False):
                    fused.PRETEST = pre
                    torch.manual_seed(77)
                    _lib.reset_stats()
                    # (3e6 draws: a workgroup takes 1536 of them, one and a half of its pre-test chunks)
                    res = _call(ml, name, P=P, N=3_000_000 if (parallel and P == 3.3 and name in ("TTP", "SEB")) else 1_000_000,
                                parallel=parallel, cc=CC, filt="J", star=star)
                    out.append((res, _lib.STATS["rows"]))
                (a, na), (b, nb) = out
                assert na == nb and na > 0, (name, P, parallel, na, nb)
                da = a if isinstance(a, tuple) else (a,)
                db = b if isinstance(b, tuple) else (b,)
                for x, y in zip(da, db):
                    for k in y:
                        assert np.array_equal(np.asarray(x[k]), np.asarray(y[k]), equal_nan=True), (name, P, parallel, k)
    finally:
        fused.TABLE_ROWS, fused.PRETEST = saved_rows, saved_pre
        triceratops_amd.set_sampling("numpy")


@pytest.mark.parametrize("name", ["STP", "SEB"])
def test_missing_limb_darkening_cell_raises_with_and_without_the_pretest(name):
    """lnZ_STP / lnZ_SEB look up the limb-darkening cell of EVERY draw's companion and raise when the Claret grid
    lacks it (marginal_likelihoods.py:945-972: .item() on an empty selection), whether the draw transits or not.
    The pre-test must not narrow that to the draws it lets through: blanking the cells of the companion lattice
    one at a time, the same cells raise with the pre-test on and off -- at N = 400 about half of the cells that any
    draw hits are hit by no transiting draw."""
    import triceratops_amd
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    triceratops_amd.set_sampling("device")
    saved_rows, saved_pre = fused.TABLE_ROWS, fused.PRETEST
    fused.TABLE_ROWS = 1
    try:
        torch.manual_seed(3)
        _call(ml, name, P=3.3, N=400, parallel=True, cc=None, filt="TESS")     # fills the lattice cache
        cap = 10000 if name == "STP" else 13000
        lut, n_lut = [v for k, v in fused._lut_cache.items() if k[0] == "TESS" and k[1] == 0.0 and k[2] == cap][0]
        clean = lut.clone()
        raised = {True: set(), False: set()}
        for cell in range(n_lut):
            if bool(torch.isnan(clean[cell])):
                continue
            lut.copy_(clean)
            lut[cell] = float("nan")
            for pre in (True, False):
                fused.PRETEST = pre
                torch.manual_seed(3)
                try:
                    _call(ml, name, P=3.3, N=400, parallel=True, cc=None, filt="TESS")
                except ValueError:
                    raised[pre].add(cell)
        lut.copy_(clean)
        assert raised[True] == raised[False]
        assert len(raised[False]) >= 3                      # the draws do spread over several cells
    finally:
        fused.TABLE_ROWS, fused.PRETEST = saved_rows, saved_pre
        fused._lut_cache.clear()
        triceratops_amd.set_sampling("numpy")


def _philox4x32_10(counter, key):
    """Random123's Philox4x32-10 (known answer: zero counter and key -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8)"""
    M = 0xffffffff
    c, (k0, k1) = list(counter), key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & M, (p0 >> 32) ^ c[3] ^ k1, p0 & M]
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    return c


def test_in_kernel_random_numbers_are_philox_blocks():
    """the uniforms a draw used (dumped) against a Python restatement of the generator: key = the call's seed,
    counter = (draw index, slot group, sub-draw); two 53-bit uniforms per block, numpy's construction"""
    assert _philox4x32_10([0, 0, 0, 0], (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    import triceratops_amd
    from triceratops_amd import fused
    from triceratops_amd import marginal_likelihoods as ml
    triceratops_amd.set_sampling("device")
    fused.DUMP = []
    try:
        torch.manual_seed(5)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())     # what fused draws for the call
        torch.manual_seed(5)
        _call(ml, "TEB", P=[2.5, 4.0], N=5000, parallel=True, cc=None, filt="TESS")
        d = fused.DUMP[0]["dump"].cpu().numpy()
    finally:
        fused.DUMP = None
        triceratops_amd.set_sampling("numpy")
    for i in (0, 1, 63, 64, 4999):
        def pair(group):
            r = _philox4x32_10([i, 0, 16 + group, 0], (seed & 0xffffffff, seed >> 32))
            return (((r[0] >> 5) * 67108864 + (r[1] >> 6)) / 9007199254740992.0,
                    ((r[2] >> 5) * 67108864 + (r[3] >> 6)) / 9007199254740992.0)
        q, inc = pair(0)            # rows of the dump: 0 P, 3 inc, 4 q, 5 ecc, 6 argp
        ecc, argp = pair(1)
        _, P = pair(2)
        assert (d[4, i], d[3, i], d[5, i], d[6, i], d[0, i]) == (q, inc, ecc, argp, P), i
