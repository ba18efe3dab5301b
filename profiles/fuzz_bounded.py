"""Randomised stress of the bounded evaluation at the level of whole calc_probs runs: bounded (mode 2, the default)
against the full evaluation (mode 0) on the same seeds, with the chi^2 arrays poisoned before every call
(trx_set_debug_poison: a row no pass wrote reads as a perfect fit).  Random light-curve length (batched and
one-row-per-wave variants, both sides of every threshold), time span, noise, signal strength (none / weak / strong /
a signal deeper than any model can be), N (so that the masked counts cross the rows-per-wave rules), stream count.
Same best draws, |lnZ difference| <= 1e-12 max(1, |lnZ|), FPP / NFPP to 1e-12 + 4 x the largest lnZ difference.
    python profiles/fuzz_bounded.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import triceratops_amd
from triceratops_amd import _lib, sharding, synth

GOLD = os.path.join(ROOT, "tests", "golden")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
triceratops_amd.set_sampling("device")
sharding.per_unit_seed = True
L = _lib.lib()
L.trx_set_debug_poison(1)
t_start, n_cfg, worst, fails = time.time(), 0, 0.0, []
stats = {}
while time.time() - t_start < budget:
    n_time = int(rng.choice([20, 33, 47, 48, 50, 63, 64, 65, 100, 137, 200, 300, 319, 320, 321, 478, 640, 769, 1000, 1500]))
    N = int(rng.choice([3000, 20_000, 33_000, 60_000, 130_000, 290_000, 330_000, 420_000, 560_000, 1_000_000]))
    if n_time >= 1000 and N > 420_000:
        N = 290_000
    kind = str(rng.choice(["signal", "noise", "scaled", "deep", "shifted", "quiet"]))
    streams = int(rng.choice([1, 2, 3, 6]))
    n_tois = int(rng.choice([1, 2]))
    toi_seed = int(rng.integers(1 << 30))
    run_seed = int(rng.integers(1 << 30))
    wide = rng.random() < 0.3             # a wider time span: most stamps out of every window
    got = {}
    try:
        for mode in (0, 2):
            L.trx_set_bounded_evaluation(mode)
            np.random.seed(run_seed)
            torch.manual_seed(run_seed)
            jobs = synth.toi_jobs(n_tois, n_time=n_time, N=N, seed=toi_seed, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                                  contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv") if toi_seed & 1 else None)
            r2 = np.random.default_rng(toi_seed + 1)
            for _, kw in jobs:
                t, f, s = kw["time"], kw["flux_0"], kw["flux_err_0"]
                if kind == "noise":              # no draw stands out
                    kw["flux_0"] = 1.0 + r2.normal(0.0, s, t.size)
                elif kind == "scaled":           # weaker or stronger signal at the same noise
                    kw["flux_0"] = 1.0 + (f - 1.0) * float(r2.choice([0.2, 0.5, 3.0]))
                elif kind == "deep":             # deeper than any planet of the prior can be
                    kw["flux_0"] = np.where(np.abs(t) < 0.03, f - 0.2, f)
                elif kind == "shifted":          # the signal where few draws have their window
                    kw["flux_0"] = np.roll(f, t.size // 3)
                elif kind == "quiet":            # error bars ten times too small: every row far from every other
                    kw["flux_err_0"] = s * 0.1
                if wide:
                    kw["time"] = t * 2.0
            sharding.streams = streams
            got[mode] = triceratops_amd.calc_probs_many(jobs)
        for x, z in zip(got[0], got[2]):
            fin = np.isfinite(x.lnZ)
            ok = np.array_equal(fin, np.isfinite(z.lnZ))
            d = float((np.abs(z.lnZ[fin] - x.lnZ[fin]) / np.maximum(1.0, np.abs(x.lnZ[fin]))).max()) if ok and fin.any() else 0.0
            dabs = float(np.abs(z.lnZ[fin] - x.lnZ[fin]).max()) if ok and fin.any() else 0.0
            worst = max(worst, d)
            why = "" if ok else "finite pattern "
            if d > 1e-12:
                why += "lnZ "
            # (error bars ten times too small put lnZ at -1e4 .. -1e6: 1e-14 relative is 1e-9 absolute, and a
            # probability moves by that much)
            if max(abs(float(x.FPP) - float(z.FPP)), abs(float(x.NFPP) - float(z.NFPP))) > 1e-12 + 4.0 * dabs:
                why += "FPP "
            for c in ("P_orb", "inc", "R_p", "ecc", "w", "M_EB", "R_EB"):
                if not np.array_equal(x.probs[c].values, z.probs[c].values, equal_nan=True):
                    why += c + " "
            if why:
                fails.append((n_time, N, kind + ("+wide" if wide else ""), streams, n_tois, toi_seed, run_seed, "%s(%.3g)" % (why, d)))
    except Exception as exc:                      # noqa: BLE001  (a fuzz reports, it does not stop)
        fails.append((n_time, N, kind + ("+wide" if wide else ""), streams, n_tois, toi_seed, run_seed, "EXC " + repr(exc)))
    n_cfg += 1
    stats[kind] = stats.get(kind, 0) + 1
L.trx_set_debug_poison(0)
L.trx_set_bounded_evaluation(2)
print("fuzz_bounded seed %d: %d configurations in %.0f s (%s), worst lnZ difference relative to max(1, |lnZ|) %.2e, %d failures"
      % (seed, n_cfg, time.time() - t_start, ", ".join("%s %d" % kv for kv in sorted(stats.items())), worst, len(fails)))
import collections
by = collections.Counter((f[0], f[2], "exception" if f[7].startswith("EXC") else f[7].split("(")[0]) for f in fails)
for key, cnt in sorted(by.items()):
    print("   %5d failures: n_time %d, %s, %s" % ((cnt,) + key))
shown = set()
for f in fails:
    key = (f[0], f[7][:40])
    if key in shown or len(shown) >= 30:
        continue
    shown.add(key)
    print("   FAIL n_time %d N %d %s streams %d tois %d toi_seed %d run_seed %d: %s" % f)
sys.exit(1 if fails else 0)
