"""The ten lnZ_* of calc_probs on the fused per-draw kernel (trx_draw_scenario, include/trx.h).

tests/torch_pipeline.py (round 1's device path) expresses a scenario as ~350 elementwise torch launches (samplers, splines,
priors, masks); here the whole draw -> derive -> mask -> prior chain of a scenario is ONE HIP
kernel over the N draws, fed with staged random numbers in the reference's draw order (the RNG
classes of device_pipeline: torch's device generator, or numpy's global stream), followed by the
compaction (nonzero + one index_select), the likelihood / log-mean-exp kernels and the best-fit
table.  About 15 launches per scenario branch instead of ~360.

Host work per call: the constants of the broken power laws (priors.py:16-383), the Moe &
Di Stefano rate constants (priors.py:601-660) and table pointers -- nothing per draw.
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib, funcs
from . import device_pipeline as dp
from . import marginal_likelihoods as ml
from ._lib import FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K, MODEL_EB, MODEL_EB_TWIN, MODEL_TP
from .constants import G, Msun, Rsun, pi

F64 = torch.float64
N_BEST = ml.N_BEST
# rows of the best-fit table a call returns.  The reference's lnZ_* return the 100 best draws but
# calc_probs reads only the best one (triceratops.py:809-823); run_units asks for 1 row while it
# evaluates the units of a calc_probs with the device generator, which turns the top-101 selection
# (a device sort, ~15 launches per branch) into one argmin.  Direct lnZ_* calls keep 100.
TABLE_ROWS = N_BEST
TABLE_MAX_ROWS = 127       # TRX_TABLE_MAX_ROWS (include/trx.h)
# True: with the device generator (set_sampling("device")) the random numbers are made inside the
# draw kernel (Philox4x32-10 keyed by one 62-bit seed per call taken from torch's CPU generator, so
# torch.manual_seed reproduces a run); False: torch's device generator fills staged arrays
PHILOX = True
_tls = threading.local()


def set_thread_seed(seed):
    """key of the calls this thread makes next (sharding.run_units: one seed per work unit, so a
    unit's draws do not depend on which rank or thread evaluates it); None = take the key from
    torch's CPU generator"""
    _tls.seed, _tls.count = seed, 0


def threadable():
    """True when lnZ_* calls may run side by side on several host threads: device sampling with
    the kernel's own random numbers (no global generator is consumed)"""
    return ml._sampling["mode"] == "device" and PHILOX and isinstance(dp.RNG, dp.TorchRng)


def staged_native():
    """True when lnZ_* calls under calc_probs go to the library's chain with STAGED random numbers: numpy's global
    stream consumed on the calling thread in the reference's order (set_sampling("numpy-device")).  One host thread
    only; the calls can still be deferred and dealt to streams like the device generator's."""
    return ml._sampling["mode"] == "numpy-device" and NATIVE and isinstance(dp.RNG, dp.NumpyStreamRng)


def _mix(seed, count):
    """splitmix64 of (seed, count): a 62-bit Philox key"""
    z = (seed * 0x9E3779B97F4A7C15 + count * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) >> 2


DUMP = None        # tests: a list that receives the [9][N] tensor of random numbers every call used
HOST_TARGET, HOST_COMPANION, HOST_FIELD = 0, 1, 2
COMP_NONE, COMP_BOUND, COMP_FIELD = 0, 1, 2
PRIOR_NONE, PRIOR_BOUND_TP, PRIOR_BOUND_EB, PRIOR_FIELD = 0, 1, 2, 3
MAX_KNOTS, N_SPLINES, MAX_CC, MAX_LUT = 16, 6, 256, 160
SPLINE_DOUBLES = 1 + 5 * MAX_KNOTS

_d3 = ctypes.c_double * 3
_vp = ctypes.c_void_p


class PowerLaw(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int), ("ones", ctypes.c_int), ("norm", ctypes.c_double),
                ("hi", _d3), ("lo", _d3), ("cum", _d3), ("p1", _d3), ("amp", _d3), ("base", _d3),
                ("ip", _d3)]


class DrawArgs(ctypes.Structure):
    _fields_ = ([("N", ctypes.c_long)]
                + [(k, ctypes.c_int) for k in ("planet", "host", "comp", "prior", "parallel", "flat",
                                               "use_cc", "n_cc", "n_lut")]
                + [(k, ctypes.c_double) for k in ("P_lo", "P_hi", "M_s", "R_s", "Teff", "u1", "u2",
                                                  "ecc_pow", "teff_cap", "f0_tess", "f0_band",
                                                  "dist_pc", "kepler_c", "f1", "f2", "f3", "t2", "t3",
                                                  "t4", "t5", "bg_const", "bg_amp")]
                + [(k, PowerLaw) for k in ("law_rp_hi", "law_rp_lo", "law_q", "law_qc")]
                + [(k, _vp) for k in ("splines", "cc_seps", "cc_cons", "lut", "f_mass", "f_radius",
                                      "f_teff", "f_logg", "f_fr", "f_delta", "f_frband", "f_u1", "f_u2",
                                      "uP", "uQc", "uRp", "uInc", "uQ", "uEcc", "uW", "ecc_in", "qc_in",
                                      "idx", "cols", "mask", "mask_twin", "lnprior", "flag")]
                + [("use_philox", ctypes.c_int), ("range_P", ctypes.c_int), ("n_field_draw", ctypes.c_long),
                   ("pretest", ctypes.c_int), ("seed", ctypes.c_ulonglong), ("dump", _vp),
                   ("sep_in", _vp), ("dm_out", _vp)])


class ScenarioArgs(ctypes.Structure):
    """trx_scenario_args (include/trx.h)"""
    _fields_ = [("draw", ctypes.POINTER(DrawArgs)), ("time", _vp), ("flux", _vp),
                ("n_time", ctypes.c_int), ("nsupersample", ctypes.c_int),
                ("sigma", ctypes.c_double), ("lnsigma", ctypes.c_double), ("exptime", ctypes.c_double),
                ("flags", ctypes.c_int), ("want_prior", ctypes.c_int),
                ("out", ctypes.POINTER(ctypes.c_double)), ("out_flag", ctypes.POINTER(ctypes.c_int)),
                ("table_rows", ctypes.c_int), ("table", _vp)]


def _table_branch(K):
    """TRX_TABLE_BRANCH(K): doubles of one branch's table (include/trx.h)"""
    return 15 * (K + 1)


SCENARIO_OUT = 18      # TRX_SCENARIO_OUT
SCEN_TIES, SCEN_STATUS = 16, 17      # record slots: rows holding the smallest chi^2; 1 = a row was never written
# True: a lnZ_* call that only has to return its best draw (calc_probs: TABLE_ROWS == 1, device
# generator) is ONE library call, trx_scenario_evidence; False: the chain of torch operators around
# trx_draw_scenario / trx_lnz_scenario below (the path of the 100-row table; kept as cross-check)
NATIVE = os.environ.get("TRX_NATIVE", "1") != "0"      # TRX_NATIVE=0: A/B runs of whole programs
# the draw kernel's fp32 pre-test of the geometry (csrc/trx_draw.hip, may_transit): the fp64 mask is evaluated
# only for the draws it lets through.  Same masks; TRX_PRETEST=0 / PRETEST = False evaluates every draw (tests)
PRETEST = os.environ.get("TRX_PRETEST", "1") != "0"


def _fn_scenario():
    L = _lib.lib()
    if not getattr(L, "trx_bound_scenario", False):          # (per library: tests switch to the testing build and back)
        _fn()
        L.trx_scenario_evidence.restype = ctypes.c_int
        L.trx_scenario_evidence.argtypes = [ctypes.POINTER(ScenarioArgs), _vp]
        L.trx_scenario_enqueue.restype = ctypes.c_int
        L.trx_scenario_enqueue.argtypes = [ctypes.POINTER(ScenarioArgs), _vp, _vp]
        L.trx_star_enqueue.restype = ctypes.c_int
        L.trx_star_enqueue.argtypes = [ctypes.POINTER(ScenarioArgs), ctypes.c_int, ctypes.POINTER(_vp),
                                       ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int)]
        L.trx_scenario_args_size.restype = ctypes.c_size_t
        if L.trx_scenario_args_size() != ctypes.sizeof(ScenarioArgs):
            raise _lib.TrxError("trx_scenario_args layout mismatch: library %d bytes, binding %d"
                                % (L.trx_scenario_args_size(), ctypes.sizeof(ScenarioArgs)))
        L.trx_bound_scenario = True
    return L.trx_scenario_enqueue


# ---------------------------------------------------------------------------------------
# Calls in flight.  trx_scenario_enqueue never waits for the device, so the lnZ_* calls of a
# calc_probs can all be enqueued (on a few streams, from one host thread) before anything is read
# back: between begin_deferred() and end_deferred() a native call returns a Pending instead of its
# result dicts, and sharding.run_units resolves them after one synchronisation per stream.
RECORD = 2 * SCENARIO_OUT + 1      # doubles per call: two branch records + the limb-darkening flag


class Pending:
    """one trx_scenario_enqueue call whose record has not been read yet"""

    def __init__(self, scen, out, stream, keep, ncol, n_time, is_host=False, table=None, table_rows=0):
        self.scen, self.out, self.stream, self.keep, self.ncol, self.n_time = scen, out, stream, keep, ncol, n_time
        self.is_host = is_host
        self.table, self.table_rows = table, table_rows        # pinned [2][15][K + 1] block of a call with a table

    def result(self):
        """the call's result dict(s); the stream must have been synchronised"""
        rec = self.out.numpy()
        if rec[2 * SCENARIO_OUT] != 0.0:
            self.keep = None
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        planet, ncol = bool(self.scen.a.planet), self.ncol
        _check_status(rec[None, :], [planet])
        with _stats_lock:
            _lib.STATS["native_calls"] += 1
        if self.replay_for_ties(rec):
            return self.scen.run_operator_chain(self.is_host, ncol)
        self.keep = None
        res = []
        K = self.table_rows
        for b in range(1 if planet else 2):
            row = rec[b * SCENARIO_OUT:(b + 1) * SCENARIO_OUT]
            n = int(row[ncol + 1])
            with _stats_lock:
                _lib.count_launch(n, self.n_time)
            if K > 1:
                blk = self.table.numpy()[b].reshape(15, K + 1)
                res.append(self.scen._table(blk[:ncol, :K].copy(), float(row[ncol]), b == 1))
            else:
                res.append(self.scen._table(row[:ncol].reshape(ncol, 1).copy(), float(row[ncol]), b == 1))
        return res[0] if planet else (res[0], res[1])


    def replay_for_ties(self, rec):
        """The seeded numpy modes promise the reference's OWN best draw, and the reference takes it from an argsort
        that orders exact ties its own way (marginal_likelihoods.py:152: introsort, not stable); the library reports
        the first of equals and how many rows tie (record slot 16).  With a tie at the minimum the call is evaluated
        again by the operator chain, on the same staged numbers (still alive in self.keep), whose best-draw search
        reproduces numpy's order.  (Exact ties at the minimum are what flat models give: a scenario none of whose
        draws touches the data.)"""
        if not isinstance(dp.RNG, dp.NumpyStreamRng) or self.scen.philox:
            return False
        nbr = 1 if self.scen.a.planet else 2
        K = self.table_rows
        if K > 1:
            # a table of the K best draws: the reference's order needs the K + 1 smallest chi^2 to be finite and
            # strictly increasing, and more than K masked draws (the operator chain's own rule, _best below: anything
            # else goes through numpy's argsort on the host)
            for b in range(nbr):
                n = int(rec[b * SCENARIO_OUT + self.ncol + 1])
                hv = self.table.numpy()[b].reshape(15, K + 1)[14]
                if n <= K or not (np.isfinite(hv[-1]) and np.all(hv[1:] > hv[:-1])):
                    return True
            return False
        return any(rec[b * SCENARIO_OUT + SCEN_TIES] > 1.0 for b in range(nbr))


def _check_status(recs, planet):
    """raises when a record says that a masked draw's chi^2 was never written (an internal error of the passes of the
    bounded evaluation: round 4 shipped two such bugs, and a stale value read as a result gave FPP = 1)"""
    for b in range(2):
        bad = recs[:, b * SCENARIO_OUT + SCEN_STATUS] != 0.0
        if b == 1:
            bad = bad & ~np.asarray(planet, dtype=bool)
        if np.any(bad):
            raise _lib.TrxError("libtrx: %d lnZ_* call(s) of this pass report rows of their likelihood that no kernel "
                                "wrote (branch %d; record status 1) -- an internal error, the results are not usable; "
                                "triceratops_amd.set_full_evaluation(True) (TRX_FLAG_FULL_EVALUATION) evaluates every "
                                "row in one pass" % (int(bad.sum()), b))


_stats_lock = threading.Lock()


def records_to_rows(pending):
    """The calls of a pass in one go: [(unit, Pending)] -> {unit: (branches, 15) array in sharding.RECORD_COLS order
    (M_s R_s u1 u2 P_orb inc b R_p ecc argp M_EB R_EB fluxratio_EB fluxratio_comp lnZ)}, what Pending.result() +
    sharding._record give call by call (768 calls of a 64-target step: 12 ms of dict building; here a few array
    expressions).  The streams must have been synchronised."""
    if not pending:
        return {}
    recs = np.stack([p.out.numpy() for _, p in pending])                 # [calls][37]
    if np.any(recs[:, 2 * SCENARIO_OUT] != 0.0):
        raise ValueError("can only convert an array of size 1 to a Python scalar")
    planet = np.array([bool(p.scen.a.planet) for _, p in pending])
    _check_status(recs, planet)
    replay = [i for i, (_, p) in enumerate(pending) if p.replay_for_ties(recs[i])]
    if replay:
        # (seeded numpy modes only: the reference's own order among exactly tied best draws, see Pending.replay_for_ties)
        redo = {}
        for i in replay:
            k, p = pending[i]
            with torch.cuda.stream(p.stream):
                res = p.scen.run_operator_chain(p.is_host, p.ncol)
            p.stream.synchronize()
            dicts = res if isinstance(res, tuple) else (res,)
            from .sharding import RECORD_COLS
            redo[k] = np.array([[d[c] if c == "lnZ" else d[c][0] for c in RECORD_COLS] for d in dicts])
        with _stats_lock:
            _lib.STATS["native_calls"] += len(replay)
        rest = [kp for i, kp in enumerate(pending) if i not in set(replay)]
        out = records_to_rows(rest)
        out.update(redo)
        for _, p in pending:
            p.keep = None
        return out
    n_time = np.array([p.n_time for _, p in pending])
    out = {}

    def impact(sm, ecc, w, inc, Rh):
        return sm * (1 - ecc ** 2) / (1 + ecc * np.sin(w * pi / 180)) * np.cos(inc * pi / 180) / (Rh * Rsun)

    rows_total = cells_total = launches = 0
    if planet.any():
        r = recs[planet]
        rp, P, inc, sm, Rh, u1, u2, ecc, w, frc, Mh, lnz, n = (r[:, j] for j in range(13))
        z = np.zeros(r.shape[0])
        block = np.stack([Mh, Rh, u1, u2, P, inc, impact(sm, ecc, w, inc, Rh), rp, ecc, w, z, z, z, frc, lnz], axis=1)
        for row, (k, _) in zip(block, [kp for kp, pl in zip(pending, planet) if pl]):
            out[k] = row[None, :]
        rows_total += int(n.sum())
        cells_total += int((n * n_time[planet]).sum())
        launches += r.shape[0]
    if (~planet).any():
        r = recs[~planet]
        blocks = []
        for b in range(2):
            q = r[:, b * SCENARIO_OUT:(b + 1) * SCENARIO_OUT]
            rr, fr, P, inc, sm, Rh, u1, u2, ecc, w, frc, sm2, m, Mh, lnz, n = (q[:, j] for j in range(16))
            if b == 1:
                P, sm = 2 * P, sm2
            z = np.zeros(q.shape[0])
            blocks.append(np.stack([Mh, Rh, u1, u2, P, inc, impact(sm, ecc, w, inc, Rh), z, ecc, w, m, rr, fr, frc, lnz], axis=1))
            rows_total += int(n.sum())
            cells_total += int((n * n_time[~planet]).sum())
            launches += q.shape[0]
        both = np.stack(blocks, axis=1)                                   # [calls][2][15]
        for row, (k, _) in zip(both, [kp for kp, pl in zip(pending, planet) if not pl]):
            out[k] = row
    with _stats_lock:
        _lib.STATS["rows"] += rows_total
        _lib.STATS["cells"] += cells_total
        _lib.STATS["launches"] += launches
        _lib.STATS["native_calls"] += len(pending)
    for _, p in pending:
        p.keep = None
    return out


def begin_deferred(n_calls):
    """this thread's native lnZ_* calls return Pending objects until end_deferred(); n_calls bounds
    their number (one pinned block holds all their records).  The calls are not handed to the library one by
    one: they collect in a list that flush() passes on in ONE call (trx_star_enqueue) -- sharding.run_units
    flushes at the end of every star's units."""
    _tls.records = torch.empty((max(int(n_calls), 1), RECORD), dtype=F64, pin_memory=True)
    _tls.next_record = 0
    _tls.batch = []


def flush():
    """hand the calls collected since the last flush to the library: one trx_star_enqueue"""
    batch = getattr(_tls, "batch", None)
    if not batch:
        return
    _tls.batch = []
    n = len(batch)
    calls = (ScenarioArgs * n)()
    outs, sts = (_vp * n)(), (_vp * n)()
    waited = set()
    for i, (sa, out, stream, dev) in enumerate(batch):
        calls[i] = sa
        outs[i], sts[i] = out.data_ptr(), stream.cuda_stream
        if stream.cuda_stream not in waited:
            waited.add(stream.cuda_stream)
            _lib.wait_uploads(stream)
    done = ctypes.c_int(0)
    _fn_scenario()
    with torch.cuda.device(batch[0][3]):
        rc = _lib.lib().trx_star_enqueue(calls, n, outs, sts, ctypes.byref(done))
    if rc:
        raise _lib.TrxError("trx_star_enqueue failed at call %d of %d with status %d: %s"
                            % (done.value, n, rc, _lib.lib().trx_last_error().decode()))


def end_deferred():
    _tls.records = None
    _tls.batch = None


def _record_slot():
    """(pinned record of the next call, deferred?)"""
    recs = getattr(_tls, "records", None)
    if recs is not None and _tls.next_record < recs.shape[0]:
        _tls.next_record += 1
        return recs[_tls.next_record - 1], True
    one = getattr(_tls, "one_record", None)
    if one is None:
        one = _tls.one_record = torch.empty(RECORD, dtype=F64).pin_memory()
    return one, False


def _fn():
    L = _lib.lib()
    if not getattr(L, "trx_bound_draw", False):
        L.trx_draw_scenario.restype = ctypes.c_int
        L.trx_draw_scenario.argtypes = [ctypes.POINTER(DrawArgs), _vp]
        L.trx_draw_args_size.restype = ctypes.c_size_t
        if L.trx_draw_args_size() != ctypes.sizeof(DrawArgs):
            raise _lib.TrxError("trx_draw_args layout mismatch: library %d bytes, binding %d"
                                % (L.trx_draw_args_size(), ctypes.sizeof(DrawArgs)))
        L.trx_bound_draw = True
    return L.trx_draw_scenario


# ---------------------------------------------------------------------------------------
# host constants
def _law(edges, powers, amps_int, amps_inv):
    """constants of tests/torch_pipeline._invert, same Python-float arithmetic"""
    law = PowerLaw()
    ints = []
    for j, p in enumerate(powers):
        span = edges[j + 1] ** (p + 1) - edges[j] ** (p + 1)
        ints.append(span / (p + 1) if amps_int[j] is None else amps_int[j] * span / (p + 1))
    norm = 1 / sum(ints)
    law.nseg, law.ones, law.norm = len(powers), 0, norm
    cum = 0.0
    for j, p in enumerate(powers):
        upper = cum + ints[j]
        law.lo[j], law.hi[j], law.cum[j] = norm * cum, norm * upper, cum
        law.p1[j] = p + 1
        law.amp[j] = 0.0 if amps_inv[j] is None else amps_inv[j]
        law.base[j] = edges[j] ** (p + 1)
        law.ip[j] = 1 / (p + 1)
        cum = upper
    return law


def _rp_laws():
    edges = (0.5, 3.0, 6.0, 20.0)
    out = []
    for powers in ((0.0, -4.0, -0.5), (0.0, -7.0, -0.5)):
        p1, p2, p3 = powers
        A1 = edges[1] ** p1 / edges[1] ** p2
        A2 = edges[2] ** p2 / edges[2] ** p3
        out.append(_law(edges, powers, (None, A1, A2 * A1), (None, A1, A1 * A2)))
    return out


_q_law_cache = {}


def _q_law(M_s, p_hi, F_twin):
    """tests/torch_pipeline._mass_ratio (priors.py:168-383); constants of a star's calls are built once"""
    key = (float(M_s), p_hi, F_twin)
    law = _q_law_cache.get(key)
    if law is None:
        if len(_q_law_cache) > 256:
            _q_law_cache.clear()
        law = _q_law_cache[key] = _q_law_build(M_s, p_hi, F_twin)
    return law


def _q_law_build(M_s, p_hi, F_twin):
    if M_s <= 0.1:
        law = PowerLaw()
        law.ones = 1
        return law
    p1, p2 = 0.3, p_hi

    def twin_amp(lo):
        return (1 + F_twin / (1 - F_twin) * ((1.0 - lo ** (p2 + 1)) / (p2 + 1))
                / ((1.0 - 0.95 ** (p2 + 1)) / (p2 + 1)))

    if M_s >= 0.3:
        q_min = 0.1 if M_s >= 1.0 else 0.1 / M_s
        A1 = (0.3 ** p1) / (0.3 ** p2)
        A2 = twin_amp(0.3)
        return _law((q_min, 0.3, 0.95, 1.0), (p1, p2, p2), (None, A1, A2 * A1), (None, A1, A1 * A2))
    q_min = 0.1 / M_s
    A2 = twin_amp(q_min)
    return _law((q_min, 0.95, 1.0), (p2, p2), (None, A2), (None, A2))


_RP = None
_tab_cache = {}
_ldc_star_cache = {}


def _spline_table(device, band):
    """the six piecewise cubics the kernel stages in LDS, [6][1 + 5 * 16] doubles"""
    key = (device.type, device.index, band)
    if key not in _tab_cache:
        from scipy.interpolate import PPoly
        tab = np.zeros((N_SPLINES, SPLINE_DOUBLES))
        fb = funcs._flux_spl[band if band in funcs._flux_spl else "TESS"]
        for i, spl in enumerate((funcs._spl["R_hot"], funcs._spl["T_hot"], funcs._spl["R_cool"],
                                 funcs._spl["T_cool"], funcs._flux_spl["TESS"], fb)):
            pp = PPoly.from_spline(spl._eval_args)
            keep = np.diff(pp.x) > 0
            x, c = pp.x[:-1][keep], pp.c[:, keep]
            m = x.size
            assert m <= MAX_KNOTS
            tab[i, 0] = m
            tab[i, 1:1 + m] = x
            for r in range(4):
                tab[i, 1 + MAX_KNOTS * (r + 1):1 + MAX_KNOTS * (r + 1) + m] = c[r]
        _tab_cache[key] = _lib.dev(tab.ravel(), device)
    return _tab_cache[key]


_flux0_cache = {}


def _flux0(M_s, band):
    """flux_relation(M_s) as tests/torch_pipeline._flux_share forms it (the ten calls of a star ask for the
    same two or three values: kept)"""
    key = (float(M_s), band)
    v = _flux0_cache.get(key)
    if v is None:
        if len(_flux0_cache) > 256:
            _flux0_cache.clear()
        v = _flux0_cache[key] = float(10 ** funcs._flux_spl[band](np.array([M_s]))[0])
    return v


_cc_cache = {}
# Experiments only (profiles/anchor_sensitivity.py): a callable(args, kind) applied to the Moe & Di Stefano constants of a
# bound-companion prior after they are set, and the angular separation that stands in for a contrast curve
# (marginal_likelihoods.py:479-487: 2.2 arcsec)
BOUND_HOOK = None
NO_CC_SEPARATION = 2.2


def _contrast_curve(cc_file, device):
    key = (cc_file, device.type, device.index, NO_CC_SEPARATION if cc_file is None else None)
    if key not in _cc_cache:
        if cc_file is None:
            seps, cons = np.array([float(NO_CC_SEPARATION)]), np.array([1.0])
        else:
            seps, cons = funcs.file_to_contrast_curve(cc_file)
        if cons.size > MAX_CC:
            raise ValueError("contrast curve has more than %d points" % MAX_CC)
        # (the host copies and whether the contrasts are monotonic: _Scenario._replay_interp)
        _cc_cache[key] = (_lib.dev(seps, device), _lib.dev(cons, device), int(cons.size), seps, cons,
                          bool(cons.size > 1 and np.any(np.diff(cons) <= 0)))
    return _cc_cache[key]


_lut_cache = {}
_field_cache = {}      # TRILEGAL populations on the device, by (file, target magnitudes, mission, device)


def _companion_lut(mission, Z, teff_cap, device):
    key = (mission, float(Z), teff_cap, device.type, device.index)
    if key not in _lut_cache:
        tab = ml._ldc(mission)
        atZ = tab.Zs == tab.Zs[np.abs(tab.Zs - Z).argmin()]
        nT = int((teff_cap - 3500) // 250) + 1
        lut = np.full((2, nT * 4), np.nan)
        for tz, gz, a1, a2 in zip(tab.Teffs[atZ], tab.loggs[atZ], tab.u1s[atZ], tab.u2s[atZ]):
            it, ig = (tz - 3500) / 250, (gz - 3.5) / 0.5
            if 0 <= it < nT and it == int(it) and 0 <= ig < 4 and ig == int(ig):
                lut[:, int(it) * 4 + int(ig)] = (a1, a2)
        assert nT * 4 <= MAX_LUT
        _lut_cache[key] = (_lib.dev(lut.ravel(), device), nT * 4)
    return _lut_cache[key]


def _bound_constants(a, M_s, plx):
    """constants of tests/torch_pipeline._bound_rate (priors.py:601-660)"""
    if np.isnan(plx):
        plx = 0.1
    M_ref = M_s if M_s >= 1.0 else 1.0
    lm = np.log10(M_ref)
    f1 = 0.020 + 0.04 * lm + 0.07 * lm ** 2
    f2 = 0.039 + 0.07 * lm + 0.01 * lm ** 2
    f3 = 0.078 - 0.05 * lm + 0.04 * lm ** 2
    alpha, dlogP = 0.018, 0.7
    k = f2 - f1 - alpha * dlogP
    k4 = f3 - f2 - alpha * dlogP
    a.dist_pc = 1000 / plx
    a.kepler_c = (4 * pi ** 2) / (G * M_ref * Msun)
    a.f1, a.f2, a.f3 = f1, f2, f3
    a.t2 = 0.5 * (2.0 * f1 + k)
    a.t3 = 0.5 * alpha * (3.4 ** 2 - 5.4 * 3.4 + 6.8) + f2 * (3.4 - 2.0)
    a.t4 = alpha * dlogP * (5.5 - 3.4) + f2 * (5.5 - 3.4) + k4 * (0.238095 * 5.5 ** 2 - 0.952381 * 5.5 + 0.485714)
    a.t5 = f3 * (3.33333 - 17.3566 * np.exp(-0.3 * 8.0))


def _ptr(t):
    return None if t is None else t.data_ptr()


_lc_cache = {}


def _on_device(a, device):
    """device copy of a light-curve array; the ~10 lnZ_* calls of one star pass the same arrays, so
    the last few are kept (keyed by content: 100-2000 doubles hash in microseconds)"""
    if isinstance(a, torch.Tensor):
        return _lib.dev(a, device)
    a = np.ascontiguousarray(a, dtype=np.float64)
    key = (a.shape, hash(a.tobytes()), device.index)
    t = _lc_cache.get(key)
    if t is None:
        if len(_lc_cache) > 32:
            _lc_cache.clear()
        t = _lc_cache[key] = _lib.dev(a, device)
    return t


# ---------------------------------------------------------------------------------------
class _Scenario:
    """one lnZ_* call: draws, the fused kernel, the branch evidences and tables"""

    def __init__(self, time, flux, sigma, N, parallel, exptime, nsamples, mission, flatpriors):
        self.dev = _lib.compute_device()
        self.time, self.flux = _on_device(time, self.dev), _on_device(flux, self.dev)
        self.sigma, self.N = float(sigma), int(N)
        self.parallel, self.exptime, self.nsamples = bool(parallel), exptime, nsamples
        self.mission, self.flat = mission, bool(flatpriors)
        self.keep = []                       # tensors the kernel reads: alive until it has run
        a = self.a = DrawArgs()
        a.N, a.parallel, a.flat = self.N, int(self.parallel), int(self.flat)
        self.philox = PHILOX and isinstance(dp.RNG, dp.TorchRng)
        if self.philox:
            a.use_philox = 1
            ts = getattr(_tls, "seed", None)
            if ts is None:
                a.seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            else:
                a.seed = _mix(int(ts), _tls.count)
                _tls.count += 1
        global _RP
        if _RP is None:
            _RP = _rp_laws()
        a.law_rp_hi, a.law_rp_lo = _RP

    def u(self):
        """pointer to one staged array of N uniforms -- or None: the kernel draws them itself"""
        if self.philox:
            return None
        t = dp.RNG.uniform(self.N, self.dev).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def period(self, P_orb):
        a = self.a
        if type(P_orb) not in [float, int]:
            a.P_lo, a.P_hi = float(P_orb[0]), float(P_orb[-1])
            a.range_P = 1
            if self.philox:
                return 0.5 * (a.P_lo + a.P_hi)          # the expectation of the mean below
            t = dp.RNG.uniform(self.N, self.dev).contiguous()
            self.keep.append(t)
            a.uP = t.data_ptr()
            # sample_ecc is handed np.mean(P_orb) of the drawn periods (e.g. marginal_likelihoods.py:103)
            return float((a.P_lo + (a.P_hi - a.P_lo) * t).mean())
        a.P_lo = a.P_hi = float(P_orb)
        return float(P_orb)

    def target(self, M_s, R_s, Teff, Z):
        a = self.a
        a.M_s, a.R_s, a.Teff = float(M_s), float(R_s), float(Teff)
        if Z is not None:
            key = (self.mission, float(Z), float(Teff), float(M_s), float(R_s))
            uu = _ldc_star_cache.get(key)
            if uu is None:
                if len(_ldc_star_cache) > 256:
                    _ldc_star_cache.clear()
                uu = _ldc_star_cache[key] = ml._ldc(self.mission).star(Z, Teff, ml._logg(M_s, R_s))
            a.u1, a.u2 = uu
        a.f0_tess = _flux0(M_s, "TESS")
        a.law_q = _q_law(M_s, -0.5, 0.30)

    def planet_draws(self, P_mean):
        a = self.a
        a.planet = 1
        a.uRp, a.uInc = self.u(), self.u()
        if not self.philox:
            dp.RNG.discard(self.N)
            # a sampler may return a strided view (torch's Beta does): the kernel reads [N] doubles
            e = dp.RNG.beta(self.N, 0.867, 3.030, self.dev).to(F64).contiguous()
            self.keep.append(e)
            a.ecc_in = e.data_ptr()
        a.uW = self.u()

    def binary_draws(self, P_mean):
        a = self.a
        a.planet = 0
        a.uInc, a.uQ = self.u(), self.u()
        dp.RNG.discard(self.N)
        a.uEcc = self.u()
        a.ecc_pow = 1.0 / (0.2 if P_mean <= 10 else 0.6)
        a.uW = self.u()

    def bound_companion(self, M_s, molusc_file):
        a = self.a
        a.comp = COMP_BOUND
        a.law_qc = _q_law(M_s, -0.95, 0.05)
        if molusc_file is None:
            a.uQc = self.u()
        else:
            q = _lib.dev(ml._bound_companions(M_s, self.N, molusc_file), self.dev).contiguous()
            self.keep.append(q)
            a.qc_in = q.data_ptr()

    def bound_prior(self, kind, M_s, plx, cc_file, filt, molusc_file):
        a = self.a
        if molusc_file is not None:
            a.prior = PRIOR_NONE
            self.want_prior = True          # lnprior = zeros
            return
        a.prior = kind
        self.want_prior = True
        _bound_constants(a, M_s, plx)
        if BOUND_HOOK is not None:
            BOUND_HOOK(a, kind)
        self._cc(cc_file, filt, M_s)

    def _cc(self, cc_file, filt, M_s):
        a = self.a
        seps, cons, n, self.cc_host_seps, self.cc_host_cons, self.cc_nonmono = _contrast_curve(cc_file, self.dev)
        a.cc_seps, a.cc_cons, a.n_cc = seps.data_ptr(), cons.data_ptr(), n
        a.use_cc = int(cc_file is not None)
        self.band = filt if (cc_file is not None and filt in ("J", "H", "K")) else "TESS"
        a.f0_band = _flux0(M_s, self.band)

    def field(self, trilegal_fname, mags, need_ldc, cc_file, filt, M_s, hi_offset):
        """TRILEGAL population + the index draw; hi_offset = -1 for the D scenarios (sic)"""
        a = self.a
        # the D and B calls of one star share the population (the last few stars' are kept)
        key = (trilegal_fname, tuple(float(m) for m in mags), self.mission, self.dev.index)
        f = _field_cache.get(key)
        if f is None:
            if len(_field_cache) >= 8:
                _field_cache.clear()
            f = _field_cache[key] = dp._Field({"device": self.dev}, trilegal_fname, *mags, self.mission, False)
        if need_ldc:
            f.need_ldc()
        self.keep.append(f)
        a.comp = COMP_FIELD
        a.f_mass, a.f_radius, a.f_teff, a.f_logg = (f.masses.data_ptr(), f.radii.data_ptr(),
                                                    f.Teffs.data_ptr(), f.loggs.data_ptr())
        a.f_fr = f.fluxratios.data_ptr()
        if need_ldc:
            a.f_u1, a.f_u2 = f.u1.data_ptr(), f.u2.data_ptr()
        self._cc(cc_file, filt, M_s)
        band = filt if cc_file is not None else "T"
        delta = f.band_delta(band).contiguous()
        frband = f.band_fluxratio(band).contiguous()
        self.keep += [delta, frband]
        a.f_delta, a.f_frband = delta.data_ptr(), frband.data_ptr()
        a.prior = PRIOR_FIELD
        self.want_prior = True
        a.bg_amp = (f.N_comp / 0.1) * (1 / 3600) ** 2
        a.bg_const = float(np.log(a.bg_amp * 2.2 ** 2))
        self.n_field, self.hi_offset = f.N_comp, hi_offset

    def field_index(self):
        self.a.n_field_draw = self.n_field + self.hi_offset
        if self.philox:
            return
        idx = dp.RNG.randint(self.n_field + self.hi_offset, self.N, self.dev).to(torch.int64).contiguous()
        self.keep.append(idx)
        self.a.idx = idx.data_ptr()

    want_prior = False
    band = "TESS"
    cc_nonmono = False

    def _replay_interp(self, ncol):
        """The reference interpolates the contrast curve with np.interp (funcs.py:222-238), and on a curve whose contrasts
        are not monotonic -- TOI-465.01's measured curve wiggles beyond 9 mag -- np.interp returns the interval its search
        ends in, a search numpy starts from the PREVIOUS draw's interval: the reference's prior of a draw on such a
        plateau depends on the draw before it.  The draw kernel bisects.  In the seeded validation mode
        ("numpy-device": the reference's draws) the reference's own values are replayed: one pass of the draw kernel
        gives every draw's contrast (trx_draw_args.dm_out), np.interp runs over them HERE, in draw order -- numpy's search
        with numpy's memory -- and the separations go back in (sep_in).  Until round 6 this was a documented deviation
        (|d lnZ| <= 8.5e-8 on TOI-465.01's D and B scenarios, tolerance 1e-6 there); now those rows meet the 1e-8 of
        all the others.  One host synchronisation, in a mode that spends its time drawing 3e8 numpy uniforms anyway."""
        a, N, dev = self.a, self.N, self.dev
        cols = torch.empty((ncol, N), dtype=F64, device=dev)
        mask = torch.empty(N, dtype=torch.uint8, device=dev)
        mask2 = torch.empty(N, dtype=torch.uint8, device=dev) if not a.planet else None
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        dm = torch.empty(N, dtype=F64, device=dev)
        a.cols, a.mask, a.mask_twin, a.lnprior, a.flag = cols.data_ptr(), mask.data_ptr(), _ptr(mask2), None, flag.data_ptr()
        a.dm_out, a.sep_in = dm.data_ptr(), None
        with torch.cuda.device(dev):
            _lib.wait_uploads(torch.cuda.current_stream(dev))
            rc = _fn()(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.TrxError("trx_draw_scenario failed with status %d" % rc)
        x = dm.abs().cpu().numpy()
        self.sep = _lib.dev(np.interp(x, self.cc_host_cons, self.cc_host_seps), dev)
        self.keep.append(self.sep)
        a.dm_out, a.sep_in = None, self.sep.data_ptr()
        a.cols = a.mask = a.mask_twin = a.flag = None

    # -----------------------------------------------------------------------------------
    def run(self, is_host):
        a, N, dev = self.a, self.N, self.dev
        tab = _spline_table(dev, self.band)
        a.splines = tab.data_ptr()
        ncol = 11 if a.planet else 14
        if (self.cc_nonmono and a.use_cc and not self.philox and isinstance(dp.RNG, dp.NumpyStreamRng)
                and a.prior in (PRIOR_BOUND_TP, PRIOR_BOUND_EB, PRIOR_FIELD)):
            self._replay_interp(ncol)
        # calc_probs (TABLE_ROWS == 1) takes the library's own chain in BOTH device modes: with numpy's stream the staged
        # uniforms go in through trx_draw_args.uP ... uW (use_philox = 0), so a seeded "numpy-device" run is the
        # production chain -- trx_star_enqueue, bounded evaluation -- on the reference's draws
        # (tests/test_gpu_production_pin.py).  The best draw is the FIRST of equal minima (numpy's argmin); the
        # reference's argsort may order exact ties differently (the operator chain below reproduces that too).
        # A direct lnZ_* call (TABLE_ROWS = 100, marginal_likelihoods.py:152-171) takes the library's chain too, since
        # round 6: the table of the K best draws is selected and gathered on the device (trx_scenario_args.table_rows;
        # every masked draw evaluated to the end).  The operator chain below remains as the cross-check, the path of the
        # dump / trace hooks, and the replay of exact ties in the seeded numpy modes.
        if NATIVE and DUMP is None and _lib.TRACE is None and TABLE_ROWS <= TABLE_MAX_ROWS:
            return self._run_native(is_host, ncol)
        return self.run_operator_chain(is_host, ncol)

    def run_operator_chain(self, is_host, ncol):
        """trx_draw_scenario (every draw in full) + torch operators for the compaction and the best-draw table +
        trx_lnz_scenario per branch: the path of the 100-row tables of direct lnZ_* calls, and the cross-check of
        the library's own chain"""
        a, N, dev = self.a, self.N, self.dev
        cols = torch.empty((ncol, N), dtype=F64, device=dev)
        mask = torch.empty(N, dtype=torch.uint8, device=dev)
        mask2 = torch.empty(N, dtype=torch.uint8, device=dev) if not a.planet else None
        lnprior = torch.empty(N, dtype=F64, device=dev) if self.want_prior else None
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        a.cols, a.mask, a.mask_twin, a.lnprior, a.flag = (cols.data_ptr(), mask.data_ptr(), _ptr(mask2),
                                                          _ptr(lnprior), flag.data_ptr())
        if DUMP is not None:
            dump = torch.zeros((9, N), dtype=F64, device=dev)
            a.dump = dump.data_ptr()
            DUMP.append({"dump": dump, "cols": cols, "mask": mask, "mask_twin": mask2, "lnprior": lnprior})
        with torch.cuda.device(dev):
            _lib.wait_uploads(torch.cuda.current_stream(dev))
            rc = _fn()(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.TrxError("trx_draw_scenario failed with status %d" % rc)
        self.keep = []
        out = []
        flags = (FLAG_COMPANION_IS_HOST if is_host else 0) | (0 if self.parallel else FLAG_SCALAR_K)
        branches = ((MODEL_TP, mask, False),) if a.planet else ((MODEL_EB, mask, False), (MODEL_EB_TWIN, mask2, True))
        for model, m, twin in branches:
            idx = torch.nonzero(m, as_tuple=False).flatten()
            n = int(idx.numel())
            nblk = 10 if a.planet else 11
            block = cols[:nblk].index_select(1, idx)
            if twin:
                block[2] *= 2.0                                  # 2 P_orb
                block[4] = cols[11].index_select(0, idx)         # a at 2 P_orb
            lp = None if lnprior is None else lnprior.index_select(0, idx)
            h, lnz = _lib.lnz_scenario(model, flags, self.time, self.flux, self.sigma, block, self.exptime,
                                       self.nsamples, lp, N, float(np.log(self.sigma)))
            best = self._best(h, idx, n)
            out.append((best, lnz, twin))
        # one device -> host copy per branch: the N_BEST x ncol table, lnZ and (once) the flag
        res = []
        for best, lnz, twin in out:
            tabl = torch.cat([cols.index_select(1, best).reshape(-1), lnz, flag.to(F64)]).cpu().numpy()
            if tabl[-1] != 0.0:
                raise ValueError("can only convert an array of size 1 to a Python scalar")
            res.append(self._table(tabl[:-2].reshape(ncol, -1), float(tabl[-2]), twin))
        return res[0] if a.planet else (res[0], res[1])

    def _run_native(self, is_host, ncol):
        """the whole call in the library: draws, masks, compaction, likelihood, evidence, best draw --
        enqueued without a host synchronisation (trx_scenario_enqueue)"""
        a, dev = self.a, self.dev
        a.pretest = int(PRETEST)
        sa = ScenarioArgs()
        sa.draw = ctypes.pointer(a)
        sa.time, sa.flux = self.time.data_ptr(), self.flux.data_ptr()
        sa.n_time, sa.nsupersample = int(self.time.numel()), int(self.nsamples)
        sa.sigma, sa.lnsigma, sa.exptime = float(self.sigma), float(np.log(self.sigma)), float(self.exptime)
        sa.flags = ((FLAG_COMPANION_IS_HOST if is_host else 0) | (0 if self.parallel else FLAG_SCALAR_K)
                    | _lib.EXTRA_FLAGS)
        sa.want_prior = int(self.want_prior)
        out, deferred = _record_slot()
        stream = torch.cuda.current_stream(dev)
        table, K = None, int(TABLE_ROWS)
        if K > 1:
            table = torch.empty((2, _table_branch(K)), dtype=F64).pin_memory()
            sa.table_rows, sa.table = K, table.data_ptr()
        pend = Pending(self, out, stream, self.keep + [self.time, self.flux], ncol, sa.n_time, is_host, table, K if K > 1 else 0)
        self.keep = []
        if deferred and getattr(_tls, "batch", None) is not None:
            _tls.batch.append((sa, out, stream, dev))      # (sa.draw points at self.a: alive in the Pending)
            return pend
        fn = _fn_scenario()
        _lib.wait_uploads(stream)
        with torch.cuda.device(dev):
            rc = fn(ctypes.byref(sa), out.data_ptr(), stream.cuda_stream)
        if rc:
            raise _lib.TrxError("trx_scenario_enqueue failed with status %d: %s"
                                % (rc, _lib.lib().trx_last_error().decode()))
        if deferred:
            return pend
        stream.synchronize()
        return pend.result()

    def _best(self, h, idx, n):
        """indices of the N_BEST best draws (see tests/torch_pipeline._evidence for the tie rules)"""
        dev, N = self.dev, self.N
        if not isinstance(dp.RNG, dp.NumpyStreamRng):
            rows = TABLE_ROWS
            if rows == 1 and n > 0:
                return idx[torch.argmin(h).reshape(1)]
            k = min(rows, n)
            best = idx[torch.topk(h, k, largest=False, sorted=True).indices] if k else idx
            if k < rows:
                best = torch.cat([best, torch.arange(rows - k, device=dev) % max(N, 1)])
            return best
        best = None
        if n > N_BEST:
            hv, hi = torch.topk(h, N_BEST + 1, largest=False, sorted=True)
            if bool(torch.isfinite(hv[-1]) & (hv[1:] > hv[:-1]).all()):
                best = idx[hi[:N_BEST]]
        if best is None:
            lnL = np.full(N, -np.inf)
            lnL[idx.cpu().numpy()] = -0.5 * np.log(2 * pi) - np.log(self.sigma) - h.cpu().numpy()
            best = torch.as_tensor((-lnL).argsort()[:N_BEST]).to(dev)
        return best

    def _table(self, t, lnZ, twin):
        """the reference's result dict from the gathered columns (marginal_likelihoods.py:152-171)"""
        z = np.zeros(t.shape[1])
        if self.a.planet:
            rp, P, inc, sm, Rh, u1, u2, ecc, w, frc, Mh = t
            b = sm * (1 - ecc ** 2) / (1 + ecc * np.sin(w * pi / 180)) * np.cos(inc * pi / 180) / (Rh * Rsun)
            return {"M_s": Mh, "R_s": Rh, "u1": u1, "u2": u2, "P_orb": P, "inc": inc, "b": b, "R_p": rp,
                    "ecc": ecc, "argp": w, "M_EB": z, "R_EB": z.copy(), "fluxratio_EB": z.copy(),
                    "fluxratio_comp": frc, "lnZ": lnZ}
        r, fr, P, inc, sm, Rh, u1, u2, ecc, w, frc, sm2, m, Mh = t
        if twin:
            P, sm = 2 * P, sm2
        b = sm * (1 - ecc ** 2) / (1 + ecc * np.sin(w * pi / 180)) * np.cos(inc * pi / 180) / (Rh * Rsun)
        return {"M_s": Mh, "R_s": Rh, "u1": u1, "u2": u2, "P_orb": P, "inc": inc, "b": b, "R_p": z,
                "ecc": ecc, "argp": w, "M_EB": m, "R_EB": r, "fluxratio_EB": fr, "fluxratio_comp": frc,
                "lnZ": lnZ}


# ---------------------------------------------------------------------------------------
# the ten scenarios: random numbers are drawn in the reference's order (App. B of SURVEY.md)
def _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples):
    s = _Scenario(time, flux, sigma, N, parallel, exptime, nsamples, mission, flatpriors)
    P_mean = s.period(P_orb)
    s.target(M_s, R_s, Teff, Z)
    return s, P_mean


def lnZ_TTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.planet_draws(Pm)
    return s.run(False)


def lnZ_TEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.binary_draws(Pm)
    return s.run(False)


def lnZ_PTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.bound_companion(M_s, molusc_file)
    s.bound_prior(PRIOR_BOUND_TP, M_s, plx, contrast_curve_file, filt, molusc_file)
    s.planet_draws(Pm)
    return s.run(False)


def lnZ_PEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.binary_draws(Pm)
    s.bound_companion(M_s, molusc_file)
    s.bound_prior(PRIOR_BOUND_EB, M_s, plx, contrast_curve_file, filt, molusc_file)
    return s.run(False)


def _companion_host(s, Z, teff_cap):
    a = s.a
    a.host = HOST_COMPANION
    lut, n = _companion_lut(s.mission, Z, teff_cap, s.dev)
    a.lut, a.n_lut, a.teff_cap = lut.data_ptr(), n, float(teff_cap)


def lnZ_STP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, None, N, parallel, mission, flatpriors, exptime, nsamples)
    s.bound_companion(M_s, molusc_file)
    _companion_host(s, Z, 10000)
    s.bound_prior(PRIOR_BOUND_TP, M_s, plx, contrast_curve_file, filt, molusc_file)
    s.planet_draws(Pm)
    return s.run(True)


def lnZ_SEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, None, N, parallel, mission, flatpriors, exptime, nsamples)
    s.binary_draws(Pm)
    s.bound_companion(M_s, molusc_file)
    _companion_host(s, Z, 13000)
    s.bound_prior(PRIOR_BOUND_EB, M_s, plx, contrast_curve_file, filt, molusc_file)
    return s.run(True)


def lnZ_DTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.field(trilegal_fname, (Tmag, Jmag, Hmag, Kmag), False, contrast_curve_file, filt, M_s, -1)
    s.field_index()
    s.planet_draws(Pm)
    return s.run(False)


def lnZ_DEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N, parallel, mission, flatpriors, exptime, nsamples)
    s.binary_draws(Pm)
    s.field(trilegal_fname, (Tmag, Jmag, Hmag, Kmag), False, contrast_curve_file, filt, M_s, -1)
    s.field_index()
    return s.run(False)


def lnZ_BTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, None, N, parallel, mission, flatpriors, exptime, nsamples)
    s.field(trilegal_fname, (Tmag, Jmag, Hmag, Kmag), True, contrast_curve_file, filt, M_s, 0)
    s.a.host = HOST_FIELD
    s.field_index()
    s.planet_draws(Pm)
    return s.run(True)


def lnZ_BEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    s, Pm = _start(time, flux, sigma, P_orb, M_s, R_s, Teff, None, N, parallel, mission, flatpriors, exptime, nsamples)
    a = s.a
    a.planet = 0
    a.uInc, a.uQ = s.u(), s.u()
    dp.RNG.discard(s.N)                 # companion mass ratios: drawn and unused (:2089)
    dp.RNG.discard(s.N)                 # sample_ecc's own uniforms
    a.uEcc = s.u()
    a.ecc_pow = 1.0 / (0.2 if Pm <= 10 else 0.6)
    a.uW = s.u()
    s.field(trilegal_fname, (Tmag, Jmag, Hmag, Kmag), True, contrast_curve_file, filt, M_s, 0)
    a.host = HOST_FIELD
    s.field_index()
    return s.run(True)
