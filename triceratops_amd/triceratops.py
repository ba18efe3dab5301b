"""`target`: the calc_probs() driver of the reference, on the MI355X kernels.

Mirrors the compute half of triceratops/triceratops.py `class target`:
  calc_depths  (triceratops.py:559-671)   aperture flux ratios and per-star transit depths
  calc_probs   (triceratops.py:673-1485)  scenario loop, probability table, FPP / NFPP
with the same argument lists and the same result attributes (.probs .lnZ .FPP .NFPP
.FPP_degenerate .star_num .u1 .u2 .fluxratio_EB .fluxratio_comp).

and the data half of plot_fits:
  fit_curves (triceratops.py:1502-1597)   best-fit model light curve per scenario

Out of scope: the catalogue / cut-out / TRILEGAL web queries of
__init__, plot_field, the matplotlib figure of plot_fits and the star-table edits (plain pandas: edit
`target.stars` directly).  A `target` here is built from a ready star table (and, for
calc_depths, pixel coordinates); the TRILEGAL population is a local csv (`trilegal_fname`).

With torch.distributed initialised (one process per GPU, RCCL) calc_probs shards the
(star, lnZ_* call) units over the ranks and finishes with ONE all_gather of the per-scenario
results; see triceratops_amd/sharding.py.
"""
import warnings

import numpy as np
from pandas import DataFrame
from scipy.special import ndtr

from . import _lib
from ._numerics import _normalize_probabilities
from .constants import G, Msun, pi
from .funcs import renorm_flux
from .marginal_likelihoods import *  # noqa: F401,F403  (reference re-exports the lnZ_* names)
from .marginal_likelihoods import (lnZ_BEB, lnZ_BTP, lnZ_DEB, lnZ_DTP, lnZ_PEB, lnZ_PTP, lnZ_SEB,
                                   lnZ_STP, lnZ_TEB, lnZ_TTP)
from . import sharding

_COLS = ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB", "R_EB",
         "fluxratio_EB", "fluxratio_comp")

# (drop key, scenario names, first row index, star_num) of the nine target-star calls, in the
# reference's order (triceratops.py:784-1340)
_TARGET_CALLS = (
    ("TP", ("TP",), 0, 1), ("EB", ("EB", "EBx2P"), 1, 1),
    ("PTP", ("PTP",), 3, 1), ("PEB", ("PEB", "PEBx2P"), 4, 1),
    ("STP", ("STP",), 6, 2), ("SEB", ("SEB", "SEBx2P"), 7, 2),
    ("DTP", ("DTP",), 9, 1), ("DEB", ("DEB", "DEBx2P"), 10, 1),
    ("BTP", ("BTP",), 12, 2), ("BEB", ("BEB", "BEBx2P"), 13, 2),
)


class target:
    def __init__(self, ID: int, sectors=None, search_radius: int = 10, mission: str = "TESS",
                 lightkurve_cache_dir=None, trilegal_fname=None, ra: float = None,
                 dec: float = None, verify_ssl: bool = True, stars: DataFrame = None,
                 pix_coords=None):
        """ID, sectors, search_radius, mission, trilegal_fname as in the reference; `stars` is the
        table the reference builds from the TIC (columns ID Tmag Jmag Hmag Kmag ra dec mass rad
        Teff plx [sep PA fluxratio tdepth]); `pix_coords` the per-sector pixel positions of those
        stars (list of (n_stars, 2) arrays), needed by calc_depths only."""
        if mission != "TESS" and mission != "Kepler" and mission != "K2":
            raise ValueError("Introduced invalid mission: " + mission)
        if stars is None:
            raise NotImplementedError(
                "triceratops_amd.target needs a ready `stars` table: the MAST/TIC/TessCut/TRILEGAL "
                "queries of the reference constructor are outside the accelerated path")
        self.ID = ID
        self.mission = mission
        self.sectors = sectors
        self.search_radius = search_radius
        self.N_pix = 2 * search_radius + 2
        self.trilegal_fname = trilegal_fname
        self.trilegal_url = None
        self.stars = stars.reset_index(drop=True)
        self.pix_coords = pix_coords

    # -----------------------------------------------------------------------------------
    def calc_depths(self, tdepth: float, all_ap_pixels=None):
        """Flux share of every star in the extraction apertures (circular Gaussian PSF,
        sigma = 0.75 px, closed-form pixel integrals) and the transit depth each star would
        need to produce the observed depth `tdepth` (fractional, like the reference)."""
        if self.pix_coords is None:
            raise ValueError("calc_depths needs pix_coords")
        if all_ap_pixels is None:
            print("No apertures provided, assuming 5x5 centered on target.")
            all_ap_pixels = []
            for coords in self.pix_coords:
                c = np.round(coords[0])
                xs = np.arange(c[0] - 2, c[0] + 3, 1)
                ys = np.arange(c[1] - 2, c[1] + 3, 1)
                all_ap_pixels.append(np.array([np.repeat(xs, 5), np.tile(ys, 5)]).T)
        sigma = 0.75
        Tmag = self.stars.Tmag.values
        amp = 10 ** ((np.min(Tmag) - Tmag) / 2.5)
        ratios = np.zeros([len(all_ap_pixels), len(self.stars)])
        for k, ap in enumerate(all_ap_pixels):
            px = np.array(ap)
            mu = np.asarray(self.pix_coords[k])
            wx = (ndtr((px[:, None, 0] + 0.5 - mu[None, :, 0]) / sigma)
                  - ndtr((px[:, None, 0] - 0.5 - mu[None, :, 0]) / sigma))
            wy = (ndtr((px[:, None, 1] + 0.5 - mu[None, :, 1]) / sigma)
                  - ndtr((px[:, None, 1] - 0.5 - mu[None, :, 1]) / sigma))
            rel = amp * np.sum(wx * wy, axis=0)
            ratios[k, :] = rel / np.sum(rel)
        flux_ratios = np.mean(ratios, axis=0)
        self.stars["fluxratio"] = flux_ratios
        tdepths = np.zeros(len(self.stars))
        nz = flux_ratios != 0
        tdepths[nz] = 1 - (flux_ratios[nz] - tdepth) / flux_ratios[nz]
        tdepths[tdepths > 1] = 0
        self.stars["tdepth"] = tdepths
        filtered = self.stars[self.stars["tdepth"] > 0]
        for i, ID in enumerate(filtered["ID"].values):
            vals = [filtered[c].values[i] for c in ("mass", "rad", "Teff")]
            if i == 0:
                vals.append(filtered["plx"].values[i])
            if np.any(np.isnan(np.array(vals, dtype=float))):
                print("WARNING: " + str(ID) + " is missing stellar properties"
                      + (" required for validation." if i == 0
                         else ". Solar values will be assumed."))
        return

    # -----------------------------------------------------------------------------------
    def _units(self, filtered, flux_0, flux_err_0, time, P_orb, contrast_curve_file, filt, N,
               parallel, drop_scenario, flatpriors, exptime, nsamples, molusc_file):
        """The independent (star, lnZ_* call) work units of one calc_probs, in the reference's
        order.  Each unit = (first row, scenario names, star_num, ID, thunk or None, drop key).
        Building the list touches nothing but the star table's columns (once each): a unit's arguments --
        the light curve renormalised to its star, the argument tuples -- are put together when its thunk
        is called, i.e. only on the rank that owns it (sharding.run_units)."""
        units = []
        ok = True
        keep, stars, column = filtered
        cache = {}

        def col(c):
            """column c of the stars that can host the signal, on first use (a rank that does not own the target never
            asks for most of them)"""
            v = cache.get(c)
            if v is None:
                v = cache[c] = column(c)[keep]
            return v
        tail = (N, parallel, self.mission, flatpriors, exptime, nsamples)
        trilegal = self.trilegal_fname
        fns = {"TP": lnZ_TTP, "EB": lnZ_TEB, "PTP": lnZ_PTP, "PEB": lnZ_PEB, "STP": lnZ_STP, "SEB": lnZ_SEB,
               "DTP": lnZ_DTP, "DEB": lnZ_DEB, "BTP": lnZ_BTP, "BEB": lnZ_BEB}

        def star_args(i, cache={}):
            """(time, flux, flux_err, P_orb, M_s, R_s, Teff) of star i, built on first use"""
            if i not in cache:
                flux, flux_err = renorm_flux(flux_0, flux_err_0, col("fluxratio")[i])
                M_s, R_s, Teff = col("mass")[i], col("rad")[i], col("Teff")[i]
                if i > 0:      # nearby star: unknown properties default to solar values
                    Teff = 5777 if np.isnan(Teff) else Teff
                    M_s = 1.0 if np.isnan(M_s) else M_s
                    R_s = 1.0 if np.isnan(R_s) else R_s
                cache[i] = (time, flux, flux_err, P_orb, M_s, R_s, Teff)
            return cache[i]

        def target_call(key):
            b, Z = star_args(0), 0.0
            if key in ("TP", "EB"):
                return fns[key](*b, Z, *tail)
            if key in ("PTP", "PEB", "STP", "SEB"):
                return fns[key](*b, Z, col("plx")[0], contrast_curve_file, filt, *tail, molusc_file)
            mags = (col("Tmag")[0], col("Jmag")[0], col("Hmag")[0], col("Kmag")[0])
            field = mags + (trilegal, contrast_curve_file, filt) + tail
            if key in ("DTP", "DEB"):
                return fns[key](*b, Z, *field)
            return fns[key](*b, *field)

        def nearby_call(i, fn):
            return fn(*star_args(i), 0.0, *tail)

        for i, ID in enumerate(stars["ID"].to_numpy()[keep]):
            if i == 0:
                if (np.isnan(col("mass")[0]) or np.isnan(col("rad")[0]) or np.isnan(col("Teff")[0])
                        or np.isnan(col("plx")[0])):
                    print("Insufficient information to validate " + str(ID)
                          + ". Please ensure a stellar mass (in M_Sun), radius (in R_Sun), Teff "
                          + "(in K), and plx (in mas) are provided in the .stars dataframe.")
                    ok = False
                    break
                for key, names, j0, snum in _TARGET_CALLS:
                    fn = None if key in drop_scenario else (lambda k=key: target_call(k))
                    units.append((j0, names, snum, ID, fn, key, 0))
            else:
                j0 = 15 + 3 * (i - 1)
                units.append((j0, ("NTP",), 1, ID, lambda i=i: nearby_call(i, lnZ_TTP), "NTP", i))
                units.append((j0 + 1, ("NEB", "NEBx2P"), 1, ID, lambda i=i: nearby_call(i, lnZ_TEB), "NEB", i))
        return units, ok

    def calc_probs(self, time, flux_0, flux_err_0: float, P_orb, contrast_curve_file: str = None,
                   filt: str = "TESS", N: int = 1000000, parallel: bool = False,
                   drop_scenario: list = [], verbose: int = 1, flatpriors: bool = False,
                   exptime: float = 0.00139, nsamples: int = 20, molusc_file: str = None):
        """Relative probability of every scenario, FPP and NFPP (same arguments as the reference).

        `parallel` selects the reference's vector-path or per-draw-loop semantics (App. C of
        SURVEY.md); both run on the GPU."""
        units, book = self._prepare(time, flux_0, flux_err_0, P_orb, contrast_curve_file, filt, N,
                                    parallel, drop_scenario, flatpriors, exptime, nsamples,
                                    molusc_file)
        self._finish(units, sharding.run_units(units, verbose=verbose, as_rows=True), book)
        return

    def _prepare(self, time, flux_0, flux_err_0, P_orb, contrast_curve_file=None, filt="TESS",
                 N=1000000, parallel=False, drop_scenario=[], flatpriors=False, exptime=0.00139,
                 nsamples=20, molusc_file=None, job=0):
        """Work units of one calc_probs (triceratops.py:673-735: NaN filter, star filter, table
        sizes) and the number of table rows.  job: index of this target in a calc_probs_many batch."""
        time = np.asarray(time, dtype=np.float64)
        flux_0 = np.asarray(flux_0, dtype=np.float64)
        keep = ~np.isnan(time) & ~np.isnan(flux_0)
        time, flux_0 = time[keep], flux_0[keep]
        # (the stars that can host the signal, triceratops.py:712; as a row mask over the table's own columns --
        # a filtered copy of the DataFrame costs more than everything else in here)
        # (the numeric columns in ONE conversion: eleven `stars[c].to_numpy()` are 30 us of pandas per target, the whole
        # table as one float64 block 7 -- 1.3 ms of a 64-target step, on every rank; a table with a non-numeric column
        # -- string IDs -- takes the columns one by one as before)
        try:
            block = self.stars.to_numpy(dtype=np.float64)
            loc = self.stars.columns.get_loc

            def column(c):
                return block[:, loc(c)]
        except (ValueError, TypeError):
            def column(c):
                return self.stars[c].to_numpy()
        keep = column("tdepth") > 0
        filtered = (keep, self.stars, column)
        n_scen = 3 * int(keep.sum()) + 12
        needs_field = not all(k in drop_scenario for k in ("DTP", "DEB", "BTP", "BEB"))
        if self.trilegal_fname is None and needs_field:
            raise ValueError("trilegal_fname is required for the D and B scenarios (the TRILEGAL "
                             "web query is outside the accelerated path); pass it to target(...) "
                             "or drop DTP, DEB, BTP and BEB")
        units, _ok = self._units(filtered, flux_0, flux_err_0, time, P_orb, contrast_curve_file,
                                 filt, N, parallel, drop_scenario, flatpriors, exptime, nsamples,
                                 molusc_file)
        # relative size of this job's units for the multi-GPU schedule, the draws per unit (stream scratch) and the
        # (job, star) the unit belongs to: the schedule deals whole jobs and whole stars first (sharding.schedule)
        weight = float(N) * max(1, time.size)
        return [u[:6] + (weight, int(N), (job, u[6])) for u in units], n_scen

    def _finish(self, units, results, n_scen, warn=True):
        """Scenario table, normalised probabilities, FPP and NFPP from the per-unit results
        (triceratops.py:1430-1485).  Plain arrays here; the `.probs` DataFrame of the reference is put
        together when it is first read (a batch of 64 targets spent as long building 64 DataFrames nobody
        had asked for yet as waiting for the GPU).  warn = False: the caller has already raised the
        reference's RuntimeWarnings for these evidences (_defer_finish)."""
        self.__dict__["_pending_finish"] = None
        targets = np.zeros(n_scen, dtype=np.dtype("i8"))
        star_num = np.zeros(n_scen, dtype=np.dtype("i8"))
        scenarios = np.zeros(n_scen, dtype=np.dtype('U6'))
        best = {c: np.zeros(n_scen) for c in _COLS}
        lnZ = np.zeros(n_scen)
        rec_tab = None
        for u, res in zip(units, results):
            j0, names, snum, ID = u[:4]
            if isinstance(res, np.ndarray):
                # (sharding.run_units(as_rows=True): the unit's (branches, 15) block of sharding.RECORD_COLS)
                if rec_tab is None:
                    rec_tab = np.zeros((n_scen, len(sharding.RECORD_COLS)))
                nb = len(names)
                rec_tab[j0:j0 + nb] = res
                targets[j0:j0 + nb], star_num[j0:j0 + nb] = ID, snum
                scenarios[j0:j0 + nb] = names
                lnZ[j0:j0 + nb] = res[:, -1]
                continue
            for off, name in enumerate(names):
                j = j0 + off
                targets[j], star_num[j], scenarios[j] = ID, snum, name
                if res is None:
                    lnZ[j] = -np.inf
                    continue
                r = res[off]
                for c in _COLS:
                    best[c][j] = r[c]
                lnZ[j] = r["lnZ"]
        if rec_tab is not None:
            for i, c in enumerate(sharding.RECORD_COLS[:-1]):
                best[c] = best[c] + rec_tab[:, i]         # (rows of dict-valued or dropped units stay as filled above)

        relative_probs, status = _normalize_probabilities(lnZ)
        if warn:
            self._warn_status(status, stacklevel=4)
        self.FPP_degenerate = status in ('anomaly', 'all_neginf')

        self._probs_columns = {
            "ID": targets, "scenario": scenarios, "M_s": best["M_s"], "R_s": best["R_s"],
            "P_orb": best["P_orb"], "inc": best["inc"], "b": best["b"], "ecc": best["ecc"],
            "w": best["argp"], "R_p": best["R_p"], "M_EB": best["M_EB"], "R_EB": best["R_EB"],
            "prob": relative_probs}
        self._probs = None
        self.lnZ = lnZ
        self.star_num = star_num
        self.u1 = best["u1"]
        self.u2 = best["u2"]
        self.fluxratio_EB = best["fluxratio_EB"]
        self.fluxratio_comp = best["fluxratio_comp"]
        prob = relative_probs
        self.FPP = 1 - (prob[0] + prob[3] + prob[9])
        self.NFPP = np.sum(prob[15:]) if len(prob) > 15 else 0.0
        return

    @staticmethod
    def _warn_status(status, stacklevel):
        """the reference's RuntimeWarnings for degenerate evidences (triceratops.py:1466-1478)"""
        if status == 'anomaly':
            warnings.warn(
                "Unexpected NaN or +inf in scenario log-evidences. This indicates a numerical "
                "anomaly unrelated to geometric exclusions. Inspect self.lnZ for diagnostics.",
                RuntimeWarning, stacklevel=stacklevel)
        elif status == 'all_neginf':
            warnings.warn(
                "All scenario log-evidences are -inf: every MC draw was geometrically invalid. "
                "FPP=1.0 reflects a failed computation, not a confident false positive. "
                "Inspect self.lnZ for diagnostics.",
                RuntimeWarning, stacklevel=stacklevel)

    # what _finish sets: a target whose table is still to be filled (calc_probs_many on several ranks) has none of them
    _RESULTS = ("lnZ", "star_num", "u1", "u2", "fluxratio_EB", "fluxratio_comp", "FPP", "NFPP", "FPP_degenerate",
                "_probs_columns", "_probs")

    def _defer_finish(self, units, results, n_scen):
        """The table of this target is filled when one of its results is first read (calc_probs_many on several
        ranks: the targets another rank evaluated).  What is kept is DATA only -- per unit (first row, names, star
        number, ID) and a copy of its own records -- not the units' closures over light curves and star tables nor
        views into the whole batch's gathered table: the target pickles and copies like any other (advisor, round 5;
        __getstate__ fills the table first), and holds on to nothing of the batch.  The reference's RuntimeWarnings
        for degenerate evidences are raised HERE, where calc_probs_many was called, not at some later read."""
        d = self.__dict__
        for name in self._RESULTS:
            d.pop(name, None)                # (results of an earlier calc_probs must not be read as this one's)
        slim = [tuple(u[:4]) for u in units]
        kept = [None if r is None else (np.array(r, copy=True) if isinstance(r, np.ndarray) else r) for r in results]
        d["_pending_finish"] = (slim, kept, n_scen)
        lnz = np.full(n_scen, 0.0)
        for u, r in zip(slim, kept):
            if r is None:
                lnz[u[0]:u[0] + len(u[1])] = -np.inf
            elif isinstance(r, np.ndarray):
                lnz[u[0]:u[0] + len(u[1])] = r[:, -1]
            else:
                lnz[u[0]:u[0] + len(u[1])] = [x["lnZ"] for x in r]
        if not np.all(np.isfinite(lnz)):      # (the common case costs one pass over ~20 numbers)
            self._warn_status(_normalize_probabilities(lnz)[1], stacklevel=4)

    def _finish_pending(self):
        d = object.__getattribute__(self, "__dict__")
        pend = d.get("_pending_finish")
        if pend is not None:
            d["_pending_finish"] = None
            self._finish(*pend, warn=False)

    def __getattr__(self, name):
        # (only reached when normal lookup fails)
        d = object.__getattribute__(self, "__dict__")
        if d.get("_pending_finish") is not None and name in type(self)._RESULTS:
            self._finish_pending()
            return object.__getattribute__(self, name)
        raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, name))

    def __getstate__(self):
        # (pickle, copy.copy and copy.deepcopy all come through here: a table still to be filled is filled first)
        self._finish_pending()
        return self.__dict__

    @property
    def probs(self):
        """the scenario table of the last calc_probs (triceratops.py:1449-1463)"""
        if getattr(self, "_probs", None) is None:
            if getattr(self, "_probs_columns", None) is None:
                raise AttributeError("'target' object has no attribute 'probs'")
            self._probs = DataFrame(self._probs_columns)
        return self._probs

    @probs.setter
    def probs(self, value):
        self._probs = value

    # -----------------------------------------------------------------------------------
    def fit_curves(self, time, flux_0, flux_err_0: float, n_model: int = 100,
                   exptime: float = 0.00139, nsamples: int = 20):
        """Best-fit model light curve of every scenario of the last calc_probs: the data half of
        the reference's plot_fits (triceratops.py:1502-1597).

        Returns (model_time, curves): model_time = linspace(min(time), max(time), n_model) and one
        dict per row of .probs {ID, scenario, flux, flux_err, model}; flux/flux_err are the data
        renormalised to that scenario's host star, model is all ones for a skipped scenario.  All
        rows of one kind (TP / EB, companion-is-host or not) go to the GPU as one parameter block,
        with the reference's scalar-path radius-ratio rule (likelihoods.py:63-66, 122-131)."""
        time = np.asarray(time, dtype=np.float64)
        flux_0 = np.asarray(flux_0, dtype=np.float64)
        live = (self.probs["ID"] != 0).values       # rows calc_probs never reached keep ID 0
        df = self.probs[live]
        star_num, u1, u2 = self.star_num[live], self.u1[live], self.u2[live]
        fr_EB, fr_comp = self.fluxratio_EB[live], self.fluxratio_comp[live]
        model_time = np.linspace(np.min(time), np.max(time), n_model)
        star_ids = self.stars["ID"].astype(str).values
        models = np.ones((len(df), n_model))
        groups = {}
        for k in range(len(df)):
            if df["M_s"].values[k] == 0.0 or not np.isfinite(df["M_s"].values[k]):
                continue
            groups.setdefault((k % 3 == 0, bool(star_num[k] != 1)), []).append(k)
        t_d = _lib.dev(model_time)
        for (is_tp, comp), rows in groups.items():
            r = np.array(rows)
            M = df["M_s"].values[r] + (0.0 if is_tp else df["M_EB"].values[r])
            P = df["P_orb"].values[r]
            a = ((G * M * Msun) / (4 * pi ** 2) * (P * 86400) ** 2) ** (1 / 3)
            common = (P, df["inc"].values[r], a, df["R_s"].values[r], u1[r], u2[r],
                      df["ecc"].values[r], df["w"].values[r], fr_comp[r])
            if is_tp:
                model, cols = _lib.MODEL_TP, (df["R_p"].values[r],) + common
            else:
                model, cols = _lib.MODEL_EB, (df["R_EB"].values[r], fr_EB[r]) + common
            flags = _lib.FLAG_SCALAR_K | (_lib.FLAG_COMPANION_IS_HOST if comp else 0)
            grid, _ = _lib.flux_grid(model, flags, t_d, _lib.dev(_lib.pack_params(model, cols, len(r))),
                                     exptime, nsamples, want_secdepth=False)
            models[r] = grid.cpu().numpy()
        curves = []
        for k in range(len(df)):
            idx = np.argwhere(star_ids == str(df["ID"].values[k]))[0, 0]
            flux, flux_err = renorm_flux(flux_0, flux_err_0, self.stars["fluxratio"].values[idx])
            curves.append({"ID": df["ID"].values[k], "scenario": df["scenario"].values[k],
                           "flux": flux, "flux_err": flux_err, "model": models[k]})
        return model_time, curves


def calc_probs_many(jobs, verbose: int = 0):
    """calc_probs for several targets at once (BASELINE config 4: many TOIs over the GPUs of a node).

    jobs: sequence of (target, kwargs) with kwargs the calc_probs arguments of that target
    (time, flux_0, flux_err_0, P_orb, ...).  The (TOI, star, lnZ_* call) units of ALL jobs form one
    list, dealt to the ranks by size (N x n_time x scenario cost) and finished with the same single
    all_gather as one calc_probs; every target then gets its own table, FPP and NFPP.  On one GPU
    without per-unit seeding this is the jobs' calc_probs calls one after the other on one random
    stream."""
    import time as _time
    t0 = _time.perf_counter()
    prepared = []
    for job, (tg, kw) in enumerate(jobs):
        kw = dict(kw)
        kw.pop("verbose", None)
        prepared.append((tg,) + tg._prepare(job=job, **kw))
    flat = [u for _, units, _ in prepared for u in units]
    t1 = _time.perf_counter()
    # (one rank: a target's table is filled as soon as its last record is in, while the GPU works on the later targets)
    finished = set()
    t_fin = [0.0]

    def job_done(job, res):
        t_a = _time.perf_counter()
        tg, units, n_scen = prepared[job]
        tg._finish(units, res, n_scen)
        finished.add(job)
        t_fin[0] += _time.perf_counter() - t_a

    results = sharding.run_units(flat, verbose=verbose, as_rows=True, job_done=job_done)
    t2 = _time.perf_counter()
    at = 0
    # Several ranks: every rank holds every record after the all_gather, and a rank fills the tables of the targets it
    # evaluated itself at once; the others' are filled when one of their results is first read (target.__getattr__) --
    # filling all 64 tables of a batch on each of eight ranks was a quarter of a rank's host path.
    many_ranks = sharding._dist() is not None
    for job, (tg, units, n_scen) in enumerate(prepared):
        if job not in finished:
            if many_ranks and job not in sharding.last_own_jobs:
                tg._defer_finish(units, results[at:at + len(units)], n_scen)
            else:
                tg._finish(units, results[at:at + len(units)], n_scen)
        at += len(units)
    # every rank lists the units of all targets (cheap: no argument is built before a unit's owner calls it)
    # and fills every target's table from the gathered records; both are a few ms for 64 targets
    sharding.timing["prepare_s"] = t1 - t0
    sharding.timing["finish_s"] = _time.perf_counter() - t2 + t_fin[0]
    return [tg for tg, _ in jobs]
