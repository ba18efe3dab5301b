"""TEST INFRASTRUCTURE (moved out of the product in round 3): the ten lnZ_* of calc_probs as an elementwise
torch expression -- round 1's device path, ~360 launches per scenario.  The product runs the same chain
in one HIP kernel (triceratops_amd/fused.py, csrc/trx_draw.hip); this module is what that kernel is checked
against (tests/test_gpu_fused.py), itself pinned to the reference's seeded goldens through numpy-stream replay
(tests/test_device_pipeline.py).  Random numbers come from triceratops_amd.device_pipeline.RNG.

Original description:
Device-resident variant of the ten lnZ_* of calc_probs.

The default path (marginal_likelihoods.py) samples the priors on the host with numpy so that a
seed reproduces the reference draw for draw; at N = 1e6 that host work is > 95 % of a
calc_probs call.  Here the whole scenario -- uniform draws, inverse-CDF samplers, stellar and
flux relations, limb-darkening gathers, companion / background priors, geometry masks, stream
compaction into the SoA parameter block, the HIP likelihood + log-mean-exp kernels, top-100
selection -- runs on the GPU; the only traffic to the host is the 100 x 14 best-fit table and
one lnZ per branch.  Same arithmetic and conventions as the host path (the formulas are those of
priors.py / funcs.py / marginal_likelihoods.py of the reference); the random numbers come from
torch's device generator (Philox), so results agree with the host path statistically, not draw
for draw.  Select with `triceratops_amd.set_sampling("device")`; torch.manual_seed seeds it.

torch is used here as array plumbing around the kernels (elementwise ops on device tensors).
"""
import numpy as np
import torch
from scipy.interpolate import PPoly

from triceratops_amd import _lib, funcs
from triceratops_amd import device_pipeline as _prod
from triceratops_amd._lib import FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K, MODEL_EB, MODEL_EB_TWIN, MODEL_TP
from triceratops_amd.constants import G, Msun, Rearth, Rsun, au, pi
from triceratops_amd import marginal_likelihoods as ml

N_BEST = ml.N_BEST
F64 = torch.float64
_COLS = ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB", "R_EB",
         "fluxratio_EB", "fluxratio_comp")


def _dev():
    return _lib.compute_device()


TorchRng, NumpyStreamRng = _prod.TorchRng, _prod.NumpyStreamRng


class _RngProxy:
    """the product's current random-number source (set_sampling swaps it)"""

    def __getattr__(self, name):
        return getattr(_prod.RNG, name)


RNG = _RngProxy()


# ---------------------------------------------------------------------------------------
# cubic splines of funcs.py as piecewise polynomials on the device
class _Spline:
    def __init__(self, spl, device):
        pp = PPoly.from_spline(spl._eval_args)
        keep = np.diff(pp.x) > 0
        self.x = torch.as_tensor(pp.x[:-1][keep], dtype=F64, device=device)
        self.c = torch.as_tensor(np.ascontiguousarray(pp.c[:, keep]), dtype=F64, device=device)

    def __call__(self, v):
        i = torch.clamp(torch.searchsorted(self.x, v, right=True) - 1, 0, self.x.numel() - 1)
        d = v - self.x[i]
        c = self.c
        return ((c[0, i] * d + c[1, i]) * d + c[2, i]) * d + c[3, i]


_cache = {}


def _tables(device):
    key = (device.type, device.index)
    if key not in _cache:
        t = {"R_hot": _Spline(funcs._spl["R_hot"], device), "T_hot": _Spline(funcs._spl["T_hot"], device),
             "R_cool": _Spline(funcs._spl["R_cool"], device), "T_cool": _Spline(funcs._spl["T_cool"], device)}
        for band in ("TESS", "J", "H", "K"):
            t["F_" + band] = _Spline(funcs._flux_spl[band], device)
        t["F_Vis"] = t["F_TESS"]
        _cache[key] = t
    return _cache[key]


def stellar_relations(M, max_R, max_T):
    """funcs.stellar_relations on device tensors (funcs.py:54-79)"""
    t = _tables(M.device)
    hot = M > 0.63
    R = torch.where(hot, t["R_hot"](M), t["R_cool"](M))
    T = torch.where(hot, t["T_hot"](M), t["T_cool"](M))
    R = torch.where(torch.isnan(M), torch.zeros_like(R), R)
    T = torch.where(torch.isnan(M), torch.zeros_like(T), T)
    R = torch.where(R > max_R, max_R, R)
    T = torch.where(T > max_T, max_T, T)
    return torch.clamp_min(R, 0.1), torch.clamp_min(T, 2800.0)


def flux_relation(M, filt="TESS"):
    return 10 ** _tables(M.device)["F_" + filt](M)


def _flux_share(M, M_s, filt="TESS"):
    f = flux_relation(M, filt)
    f0 = flux_relation(torch.tensor([M_s], dtype=F64, device=M.device), filt)
    return f / (f + f0)


def _interp(x, xp, fp):
    """np.interp for increasing xp (end values held outside the range)"""
    xp = torch.as_tensor(xp, dtype=F64, device=x.device)
    fp = torch.as_tensor(fp, dtype=F64, device=x.device)
    if xp.numel() == 1:
        return fp[0].expand_as(x).clone()
    i = torch.clamp(torch.searchsorted(xp, x, right=True) - 1, 0, xp.numel() - 2)
    w = (x - xp[i]) / (xp[i + 1] - xp[i])
    y = fp[i] + w * (fp[i + 1] - fp[i])
    y = torch.where(x <= xp[0], fp[0], y)
    return torch.where(x >= xp[-1], fp[-1], y)


# ---------------------------------------------------------------------------------------
# samplers (priors.py:16-383) as inverse CDFs on device tensors
def _invert(x, edges, powers, amps_int, amps_inv):
    ints = []
    for j, p in enumerate(powers):
        span = edges[j + 1] ** (p + 1) - edges[j] ** (p + 1)
        ints.append(span / (p + 1) if amps_int[j] is None else amps_int[j] * span / (p + 1))
    norm = 1 / sum(ints)
    out = x.clone()
    cum = 0.0
    t0 = x / norm
    for j, p in enumerate(powers):
        upper = cum + ints[j]
        sel = (x <= norm * upper) if j == 0 else ((x > norm * cum) & (x <= norm * upper))
        t = (t0 - cum) * (p + 1)
        if amps_inv[j] is not None:
            t = t / amps_inv[j]
        out = torch.where(sel, (t + edges[j] ** (p + 1)) ** (1 / (p + 1)), out)
        cum = upper
    return out


def sample_rp(x, M_s, flatpriors):
    if flatpriors:
        return x * 19.5 + 0.5
    edges = (0.5, 3.0, 6.0, 20.0)
    res = []
    for powers in ((0.0, -4.0, -0.5), (0.0, -7.0, -0.5)):
        p1, p2, p3 = powers
        A1 = edges[1] ** p1 / edges[1] ** p2
        A2 = edges[2] ** p2 / edges[2] ** p3
        res.append(_invert(x, edges, powers, (None, A1, A2 * A1), (None, A1, A1 * A2)))
    return torch.where(M_s > 0.45, res[0], res[1])


def sample_inc(x):
    return torch.arccos(1.0 - x) * 180 / np.pi


def sample_ecc(n, planet, P_orb, device):
    RNG.discard(n)                       # the reference's sample_ecc ignores its uniforms
    if planet:
        return RNG.beta(n, 0.867, 3.030, device)
    a = 0.2 if P_orb <= 10 else 0.6
    return RNG.uniform(n, device) ** (1.0 / a)


def _mass_ratio(x, M_s, p_hi, F_twin):
    if M_s <= 0.1:
        return torch.ones_like(x)
    p1, p2 = 0.3, p_hi

    def twin_amp(lo):
        return (1 + F_twin / (1 - F_twin) * ((1.0 - lo ** (p2 + 1)) / (p2 + 1))
                / ((1.0 - 0.95 ** (p2 + 1)) / (p2 + 1)))

    if M_s >= 0.3:
        q_min = 0.1 if M_s >= 1.0 else 0.1 / M_s
        A1 = (0.3 ** p1) / (0.3 ** p2)
        A2 = twin_amp(0.3)
        return _invert(x, (q_min, 0.3, 0.95, 1.0), (p1, p2, p2), (None, A1, A2 * A1), (None, A1, A1 * A2))
    q_min = 0.1 / M_s
    A2 = twin_amp(q_min)
    return _invert(x, (q_min, 0.95, 1.0), (p2, p2), (None, A2), (None, A2))


def sample_q(x, M_s):
    return _mass_ratio(x, M_s, -0.5, 0.30)


def sample_q_companion(x, M_s):
    return _mass_ratio(x, M_s, -0.95, 0.05)


# ---------------------------------------------------------------------------------------
# companion / background priors (priors.py:580-1005)
def _bound_rate(M_s, plx, delta_mags, seps_cc, cons_cc, keep_close):
    if np.isnan(plx):
        plx = 0.1
    seps = (1000 / plx) * _interp(delta_mags, cons_cc, seps_cc)
    M_ref = M_s if M_s >= 1.0 else 1.0
    lm = np.log10(M_ref)
    f1 = 0.020 + 0.04 * lm + 0.07 * lm ** 2
    f2 = 0.039 + 0.07 * lm + 0.01 * lm ** 2
    f3 = 0.078 - 0.05 * lm + 0.04 * lm ** 2
    alpha, dlogP = 0.018, 0.7
    lp = torch.log10(((4 * pi ** 2) / (G * M_ref * Msun) * (seps * au) ** 3) ** 0.5 / 86400)
    k = f2 - f1 - alpha * dlogP
    t2p = 0.5 * (lp - 1.0) * (2.0 * f1 + k * (lp - 1.0))
    t2 = 0.5 * (2.0 * f1 + k)
    t3p = 0.5 * alpha * (lp ** 2 - 5.4 * lp + 6.8) + f2 * (lp - 2.0)
    t3 = 0.5 * alpha * (3.4 ** 2 - 5.4 * 3.4 + 6.8) + f2 * (3.4 - 2.0)
    k4 = f3 - f2 - alpha * dlogP
    t4p = alpha * dlogP * (lp - 3.4) + f2 * (lp - 3.4) + k4 * (0.238095 * lp ** 2 - 0.952381 * lp + 0.485714)
    t4 = alpha * dlogP * (5.5 - 3.4) + f2 * (5.5 - 3.4) + k4 * (0.238095 * 5.5 ** 2 - 0.952381 * 5.5 + 0.485714)
    t5p = f3 * (3.33333 - 17.3566 * torch.exp(-0.3 * lp))
    t5 = f3 * (3.33333 - 17.3566 * np.exp(-0.3 * 8.0))
    z = torch.zeros_like(lp)
    if keep_close:
        f = torch.where(lp >= 8.0, z + (t2 + t3 + t4 + t5),
            torch.where(lp >= 5.5, t2 + t3 + t4 + t5p,
            torch.where(lp >= 3.4, t2 + t3 + t4p,
            torch.where(lp >= 2.0, t2 + t3p,
            torch.where(lp >= 1.0, t2p, z)))))
    else:
        f = torch.where(lp >= 8.0, z + (t4 + t5),
            torch.where(lp >= 5.5, t4 + t5p,
            torch.where(lp >= 3.4, t4p, z)))
    if M_s < 1.0:
        f = torch.clamp_min(0.65 * f + 0.35 * f * M_s, 0.0)
    return torch.log(f)


def _clip_prior(lnprior, delta_mags):
    lnprior = torch.clamp_max(lnprior, 0.0)
    return torch.where(delta_mags > 0.0, torch.full_like(lnprior, -np.inf), lnprior)


def _bound_prior(kind, M_s, plx, cc_file, fr_tess, fr_cc_fn):
    if cc_file is None:
        delta_mags = 2.5 * torch.log10(fr_tess)
        seps, cons = np.array([2.2]), np.array([1.0])
    else:
        delta_mags = 2.5 * torch.log10(fr_cc_fn())
        seps, cons = funcs.file_to_contrast_curve(cc_file)
    return _clip_prior(_bound_rate(M_s, plx, delta_mags.abs(), seps, cons, kind == "EB"), delta_mags)


# ---------------------------------------------------------------------------------------
def _sma(M_tot, P_days):
    return ((G * M_tot * Msun) / (4 * pi ** 2) * (P_days * 86400) ** 2) ** (1 / 3)


def _periods(P_orb, N, device):
    if type(P_orb) not in [float, int]:
        lo, hi = float(P_orb[0]), float(P_orb[-1])
        return lo + (hi - lo) * RNG.uniform(N, device)
    return torch.full((N,), float(P_orb), dtype=F64, device=device)


def _transits(Ptra, incs, parallel=True):
    """draws inclined enough to transit.  Vector path of the reference: inc_min = 90 where
    Ptra > 1; its per-draw loop skips such draws (`continue`)."""
    ok = Ptra <= 1.0
    inc_min = torch.where(ok, torch.arccos(torch.clamp(Ptra, -1.0, 1.0)) * 180.0 / pi,
                          torch.full_like(Ptra, 90.0))
    hit = incs >= inc_min
    return hit if parallel else (hit & ok)


def _col(v, N, device):
    return v if isinstance(v, torch.Tensor) else torch.full((N,), float(v), dtype=F64, device=device)


def _gather_rows(cols, idx, device):
    """[len(cols), len(idx)] block: one index_select over the stacked tensor columns (instead of
    one gather kernel per column) and a fill per scalar column"""
    n = int(idx.numel())
    block = torch.empty((len(cols), n), dtype=F64, device=device)
    t_rows = [i for i, c in enumerate(cols) if isinstance(c, torch.Tensor)]
    s_rows = [i for i, c in enumerate(cols) if not isinstance(c, torch.Tensor)]
    if t_rows:
        block[t_rows] = torch.stack([cols[i] for i in t_rows]).index_select(1, idx)
    for i in s_rows:      # fill kernels stay asynchronous (a host-to-device copy of the values would not)
        block[i].fill_(0.0 if cols[i] is None else float(cols[i]))
    return block


def _evidence(model, is_host, time_d, flux_d, sigma, cols, mask, lnprior, N, exptime, nsamples,
              parallel=True):
    idx = torch.nonzero(mask, as_tuple=False).flatten()
    dev = time_d.device
    n = int(idx.numel())
    block = _gather_rows(cols, idx, dev)
    lp = None if lnprior is None else lnprior[idx].contiguous()
    # the reference's per-draw loop calls the scalar lnL_* (abs(k - 1) < 1e-6 rule, 1/k secondary)
    flags = (FLAG_COMPANION_IS_HOST if is_host else 0) | (0 if parallel else FLAG_SCALAR_K)
    h, lnz = _lib.lnz_scenario(model, flags, time_d, flux_d, sigma, block, exptime, nsamples, lp, N,
                               float(np.log(sigma)))
    # best draws = smallest chi^2/2.  With torch's own generator a device top-k is all there is
    # to it.  When numpy's stream is replayed the table has to be the reference's: its order for
    # more than N_BEST finite, distinct values is the top-k's, but few surviving draws, or the
    # equal chi^2 of draws whose model is flat over the data window, come out in whatever order
    # the reference's (-lnL).argsort() gives the ties -- so that very call is made on the host.
    if not isinstance(_prod.RNG, NumpyStreamRng):
        k = min(N_BEST, n)
        best = idx[torch.topk(h, k, largest=False, sorted=True).indices] if k else idx
        if k < N_BEST:
            # fewer surviving draws than table rows: the reference fills the rest with whatever
            # -inf draws its argsort puts next; here the first draws of the block stand in for them
            pad = torch.arange(N_BEST - k, device=dev) % max(N, 1)
            best = torch.cat([best, pad])
        return best, lnz
    best = None
    if n > N_BEST:
        hv, hi = torch.topk(h, N_BEST + 1, largest=False, sorted=True)
        if bool(torch.isfinite(hv[-1]) & (hv[1:] > hv[:-1]).all()):
            best = idx[hi[:N_BEST]]
    if best is None:
        lnL = np.full(N, -np.inf)
        lnL[idx.cpu().numpy()] = -0.5 * np.log(2 * pi) - np.log(sigma) - h.cpu().numpy()
        best = torch.as_tensor((-lnL).argsort()[:N_BEST]).to(dev)
    return best, lnz


def _table(best, lnz, N, device, **cols):
    """one device->host copy: the N_BEST x 14 table and lnZ"""
    k = int(best.numel())
    tab = _gather_rows([cols[key] for key in _COLS], best, device).cpu().numpy()
    res = {}
    for i, key in enumerate(_COLS):
        col = np.zeros(N_BEST) if not isinstance(cols[key], torch.Tensor) else np.full(N_BEST, np.nan)
        if not isinstance(cols[key], torch.Tensor):
            col[:] = 0.0 if cols[key] is None else float(cols[key])
        col[:k] = tab[i]
        res[key] = col
    res["lnZ"] = float(lnz.cpu()[0])
    return res


def _planet_branch(ctx, P_orb, rps, incs, eccs, argps, a, M_host, R_host, u1, u2, fr_comp, is_host,
                   extra, lnprior):
    N, dev = ctx["N"], ctx["device"]
    sinw = torch.sin(argps * pi / 180)
    size = rps * Rearth + R_host * Rsun
    Ptra = size / a * ((1 + eccs * sinw) / (1 - eccs ** 2))
    b = a * (1 - eccs ** 2) / (1 + eccs * sinw) * torch.cos(incs * pi / 180) / (R_host * Rsun)
    mask = _transits(Ptra, incs, ctx["parallel"]) & ~(size > a * (1 - eccs))
    if extra is not None:
        mask = mask & extra
    cols = (rps, P_orb, incs, a, R_host, u1, u2, eccs, argps, 0.0 if fr_comp is None else fr_comp)
    best, lnz = _evidence(MODEL_TP, is_host, ctx["time"], ctx["flux"], ctx["sigma"], cols, mask,
                          lnprior, N, ctx["exptime"], ctx["nsamples"], ctx["parallel"])
    return _table(best, lnz, N, dev, M_s=M_host, R_s=R_host, u1=u1, u2=u2, P_orb=P_orb, inc=incs, b=b,
                  R_p=rps, ecc=eccs, argp=argps, M_EB=None, R_EB=None, fluxratio_EB=None,
                  fluxratio_comp=fr_comp)


def _binary_branches(ctx, P_orb, qs, incs, eccs, argps, masses, radii, fluxratios, M_host, R_host,
                     u1, u2, fr_comp, is_host, extra, lnprior):
    N, dev = ctx["N"], ctx["device"]
    sinw = torch.sin(argps * pi / 180)
    e_corr = (1 + eccs * sinw) / (1 - eccs ** 2)
    a = _sma(M_host + masses, P_orb)
    a_twin = _sma(M_host + masses, 2 * P_orb)
    size = radii * Rsun + R_host * Rsun
    cosi = torch.cos(incs * pi / 180)
    geo = (1 - eccs ** 2) / (1 + eccs * sinw) * cosi / (R_host * Rsun)
    out = []
    frc = 0.0 if fr_comp is None else fr_comp
    for model, per, sma, coll, qsel in (
            (MODEL_EB, P_orb, a, size > a * (1 - eccs), qs < 0.95),
            (MODEL_EB_TWIN, 2 * P_orb, a_twin, (2 * R_host * Rsun) > a_twin * (1 - eccs), qs >= 0.95)):
        mask = _transits(size / sma * e_corr, incs, ctx["parallel"]) & ~coll & qsel
        if not ctx["parallel"] and model == MODEL_EB_TWIN:
            mask = mask & (size / a * e_corr <= 1.0)    # the loop `continue`s before the twin test
        if extra is not None:
            mask = mask & extra
        cols = (radii, fluxratios, per, incs, sma, R_host, u1, u2, eccs, argps, frc)
        best, lnz = _evidence(model, is_host, ctx["time"], ctx["flux"], ctx["sigma"], cols, mask,
                              lnprior, N, ctx["exptime"], ctx["nsamples"], ctx["parallel"])
        out.append(_table(best, lnz, N, dev, M_s=M_host, R_s=R_host, u1=u1, u2=u2, P_orb=per,
                          inc=incs, b=sma * geo, R_p=None, ecc=eccs, argp=argps, M_EB=masses,
                          R_EB=radii, fluxratio_EB=fluxratios, fluxratio_comp=fr_comp))
    return out[0], out[1]


def _ctx(time, flux, sigma, N, exptime, nsamples, parallel=True):
    dev = _dev()
    return {"time": _lib.dev(time, dev), "flux": _lib.dev(flux, dev), "sigma": float(sigma), "N": int(N),
            "exptime": exptime, "nsamples": nsamples, "device": dev, "parallel": bool(parallel)}


def _rand(ctx):
    return RNG.uniform(ctx["N"], ctx["device"])


def _draw_planet(ctx, M_for_rp, P_mean, flatpriors):
    rps = sample_rp(_rand(ctx), M_for_rp, flatpriors)
    incs = sample_inc(_rand(ctx))
    eccs = sample_ecc(ctx["N"], True, P_mean, ctx["device"])
    argps = _rand(ctx) * 360
    return rps, incs, eccs, argps


def _draw_binary(ctx, M_s, P_mean):
    incs = sample_inc(_rand(ctx))
    qs = sample_q(_rand(ctx), M_s)
    eccs = sample_ecc(ctx["N"], False, P_mean, ctx["device"])
    argps = _rand(ctx) * 360
    return incs, qs, eccs, argps


def _bound_companions(ctx, M_s, molusc_file):
    if molusc_file is None:
        return sample_q_companion(_rand(ctx), M_s)
    return _lib.dev(ml._bound_companions(M_s, ctx["N"], molusc_file), ctx["device"])


def _full(ctx, v):
    return torch.full((ctx["N"],), float(v), dtype=F64, device=ctx["device"])


# ---------------------------------------------------------------------------------------
def lnZ_TTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    rps, incs, eccs, argps = _draw_planet(ctx, _full(ctx, M_s), float(P.mean()), flatpriors)
    return _planet_branch(ctx, P, rps, incs, eccs, argps, _sma(M_s, P), M_s, R_s, u1, u2, None, False,
                          None, None)


def lnZ_TEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    incs, qs, eccs, argps = _draw_binary(ctx, M_s, float(P.mean()))
    masses = qs * M_s
    radii, _ = stellar_relations(masses, _full(ctx, R_s), _full(ctx, Teff))
    return _binary_branches(ctx, P, qs, incs, eccs, argps, masses, radii, _flux_share(masses, M_s),
                            M_s, R_s, u1, u2, None, False, None, None)


def _ratio(f):
    return f / (1 - f)


def lnZ_PTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    qc = _bound_companions(ctx, M_s, molusc_file)
    mc = qc * M_s
    frc = _flux_share(mc, M_s)
    lnprior = (torch.zeros_like(frc) if molusc_file is not None else
               _bound_prior("TP", M_s, plx, contrast_curve_file, _ratio(frc),
                            lambda: _ratio(_flux_share(mc, M_s, filt))))
    rps, incs, eccs, argps = _draw_planet(ctx, _full(ctx, M_s), float(P.mean()), flatpriors)
    return _planet_branch(ctx, P, rps, incs, eccs, argps, _sma(M_s, P), M_s, R_s, u1, u2, frc, False,
                          qc != 0.0, lnprior)


def lnZ_PEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    incs, qs, eccs, argps = _draw_binary(ctx, M_s, float(P.mean()))
    qc = _bound_companions(ctx, M_s, molusc_file)
    masses = qs * M_s
    radii, _ = stellar_relations(masses, _full(ctx, R_s), _full(ctx, Teff))
    mc = qc * M_s
    frc = _flux_share(mc, M_s)
    lnprior = (torch.zeros_like(frc) if molusc_file is not None else
               _bound_prior("EB", M_s, plx, contrast_curve_file, _ratio(frc),
                            lambda: _ratio(_flux_share(mc, M_s, filt))))
    return _binary_branches(ctx, P, qs, incs, eccs, argps, masses, radii, _flux_share(masses, M_s),
                            M_s, R_s, u1, u2, frc, False, qc != 0.0, lnprior)


def _companion_host(ctx, M_s, R_s, Teff, Z, mission, molusc_file, teff_cap):
    qc = _bound_companions(ctx, M_s, molusc_file)
    mc = qc * M_s
    Rc, Tc = stellar_relations(mc, _full(ctx, R_s), _full(ctx, Teff))
    logg = torch.log10(G * (mc * Msun) / (Rc * Rsun) ** 2)
    # rounded (Teff/250, logg/0.5) lattice at the nearest Z; a draw in a cell the grid lacks raises
    # like the reference
    tab = ml._ldc(mission)
    atZ = tab.Zs == tab.Zs[np.abs(tab.Zs - Z).argmin()]
    nT = int((teff_cap - 3500) // 250) + 1
    lut = np.full((2, nT * 4), np.nan)
    for tz, gz, a1, a2 in zip(tab.Teffs[atZ], tab.loggs[atZ], tab.u1s[atZ], tab.u2s[atZ]):
        it, ig = (tz - 3500) / 250, (gz - 3.5) / 0.5
        if 0 <= it < nT and it == int(it) and 0 <= ig < 4 and ig == int(ig):
            lut[:, int(it) * 4 + int(ig)] = (a1, a2)
    lut = torch.as_tensor(lut, dtype=F64, device=ctx["device"])
    ig = torch.clamp(torch.round(logg / 0.5) * 0.5, 3.5, 5.0)
    it = torch.clamp(torch.round(Tc / 250) * 250, 3500.0, float(teff_cap))
    code = torch.nan_to_num(torch.round((it - 3500) / 250) * 4 + torch.round((ig - 3.5) / 0.5),
                            nan=0.0).long().clamp_(0, nT * 4 - 1)
    u1s, u2s = lut[0, code], lut[1, code]
    if bool(torch.isnan(u1s).any()):
        # a rounded (Teff, logg) cell the Claret grid lacks (e.g. SEB companions hotter than
        # 10000 K): the reference's `.item()` on the empty match raises, and so does the host path
        raise ValueError("can only convert an array of size 1 to a Python scalar")
    return qc, mc, Rc, Tc, _flux_share(mc, M_s), u1s, u2s


def lnZ_STP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    qc, mc, Rc, _, frc, u1s, u2s = _companion_host(ctx, M_s, R_s, Teff, Z, mission, molusc_file, 10000)
    lnprior = (torch.zeros_like(frc) if molusc_file is not None else
               _bound_prior("TP", M_s, plx, contrast_curve_file, _ratio(frc),
                            lambda: _ratio(_flux_share(mc, M_s, filt))))
    rps, incs, eccs, argps = _draw_planet(ctx, mc, float(P.mean()), flatpriors)
    return _planet_branch(ctx, P, rps, incs, eccs, argps, _sma(mc, P), mc, Rc, u1s, u2s, frc, True,
                          qc != 0.0, lnprior)


def lnZ_SEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, plx, contrast_curve_file=None, filt="TESS",
            N=1000000, parallel=False, mission="TESS", flatpriors=False, exptime=0.00139,
            nsamples=20, molusc_file=None):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    incs, qs, eccs, argps = _draw_binary(ctx, M_s, float(P.mean()))
    qc, mc, Rc, Tc, frc, u1s, u2s = _companion_host(ctx, M_s, R_s, Teff, Z, mission, molusc_file, 13000)
    masses = qs * mc
    radii, _ = stellar_relations(masses, Rc, Tc)
    fr = _flux_share(masses, M_s)
    lnprior = (torch.zeros_like(frc) if molusc_file is not None else
               _bound_prior("EB", M_s, plx, contrast_curve_file, _ratio(frc) + _ratio(fr),
                            lambda: _ratio(_flux_share(mc, M_s, filt)) + _ratio(_flux_share(masses, M_s, filt))))
    return _binary_branches(ctx, P, qs, incs, eccs, argps, masses, radii, fr, mc, Rc, u1s, u2s, frc,
                            True, qc != 0.0, lnprior)


class _Field(_prod._Field):
    """the product's device TRILEGAL population + the torch expression of the background prior"""

    def prior(self, ctx, idxs, cc_file, filt, fr_term=None, fr_term_cc=None):
        if cc_file is None:
            if fr_term is None:
                fr_term = _ratio(self.fluxratios[idxs])
            delta_mags = 2.5 * torch.log10(fr_term)
            lnprior = _full(ctx, np.log((self.N_comp / 0.1) * (1 / 3600) ** 2 * 2.2 ** 2))
        else:
            delta_mags = self.band_delta(filt)[idxs] if fr_term_cc is None else 2.5 * torch.log10(fr_term_cc)
            seps, cons = funcs.file_to_contrast_curve(cc_file)
            s = _interp(delta_mags.abs(), cons, seps)
            lnprior = torch.log((self.N_comp / 0.1) * (1 / 3600) ** 2 * s ** 2)
        return _clip_prior(lnprior, delta_mags)


def _randint(ctx, hi):
    return RNG.randint(hi, ctx["N"], ctx["device"])


def lnZ_DTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    field = _Field(ctx, trilegal_fname, Tmag, Jmag, Hmag, Kmag, mission, False)
    idxs = _randint(ctx, field.N_comp - 1)            # sic (marginal_likelihoods.py:1463)
    lnprior = field.prior(ctx, idxs, contrast_curve_file, filt)
    rps, incs, eccs, argps = _draw_planet(ctx, _full(ctx, M_s), float(P.mean()), flatpriors)
    return _planet_branch(ctx, P, rps, incs, eccs, argps, _sma(M_s, P), M_s, R_s, u1, u2,
                          field.fluxratios[idxs], False, None, lnprior)


def lnZ_DEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Z, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    u1, u2 = ml._ldc(mission).star(Z, Teff, ml._logg(M_s, R_s))
    incs, qs, eccs, argps = _draw_binary(ctx, M_s, float(P.mean()))
    masses = qs * M_s
    radii, _ = stellar_relations(masses, _full(ctx, R_s), _full(ctx, Teff))
    field = _Field(ctx, trilegal_fname, Tmag, Jmag, Hmag, Kmag, mission, False)
    idxs = _randint(ctx, field.N_comp - 1)
    lnprior = field.prior(ctx, idxs, contrast_curve_file, filt)
    return _binary_branches(ctx, P, qs, incs, eccs, argps, masses, radii, _flux_share(masses, M_s),
                            M_s, R_s, u1, u2, field.fluxratios[idxs], False, None, lnprior)


def lnZ_BTP(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    field = _Field(ctx, trilegal_fname, Tmag, Jmag, Hmag, Kmag, mission, True)
    idxs = _randint(ctx, field.N_comp)
    lnprior = field.prior(ctx, idxs, contrast_curve_file, filt)
    Mh, Rh = field.masses[idxs], field.radii[idxs]
    rps, incs, eccs, argps = _draw_planet(ctx, Mh, float(P.mean()), flatpriors)
    extra = (field.loggs[idxs] >= 3.5) & (field.Teffs[idxs] <= 10000)
    return _planet_branch(ctx, P, rps, incs, eccs, argps, _sma(Mh, P), Mh, Rh, field.u1[idxs],
                          field.u2[idxs], field.fluxratios[idxs], True, extra, lnprior)


def lnZ_BEB(time, flux, sigma, P_orb, M_s, R_s, Teff, Tmag, Jmag, Hmag, Kmag, trilegal_fname,
            contrast_curve_file=None, filt="TESS", N=1000000, parallel=False, mission="TESS",
            flatpriors=False, exptime=0.00139, nsamples=20):
    ctx = _ctx(time, flux, sigma, N, exptime, nsamples, parallel)
    P = _periods(P_orb, N, ctx["device"])
    incs = sample_inc(_rand(ctx))
    qs = sample_q(_rand(ctx), M_s)
    RNG.discard(ctx["N"])                # companion mass ratios: drawn and unused (:2089)
    eccs = sample_ecc(ctx["N"], False, float(P.mean()), ctx["device"])
    argps = _rand(ctx) * 360
    field = _Field(ctx, trilegal_fname, Tmag, Jmag, Hmag, Kmag, mission, True)
    idxs = _randint(ctx, field.N_comp)
    Mh, Rh = field.masses[idxs], field.radii[idxs]
    masses = qs * Mh
    radii, _ = stellar_relations(masses, Rh, field.Teffs[idxs])
    frc = field.fluxratios[idxs]
    fr = _flux_share(masses, M_s) * (frc / _flux_share(Mh, M_s))
    fr_term = _ratio(frc) + _ratio(fr)
    fr_term_cc = None
    if contrast_curve_file is not None:
        frc_cc = field.band_fluxratio(filt)[idxs]
        fr_cc = _flux_share(masses, M_s, filt) * (frc_cc / _flux_share(Mh, M_s, filt))
        fr_term_cc = _ratio(frc_cc) + _ratio(fr_cc)
    lnprior = field.prior(ctx, idxs, contrast_curve_file, filt, fr_term, fr_term_cc)
    extra = (field.loggs[idxs] >= 3.5) & (field.Teffs[idxs] <= 10000)
    return _binary_branches(ctx, P, qs, incs, eccs, argps, masses, radii, fr, Mh, Rh, field.u1[idxs],
                            field.u2[idxs], frc, True, extra, lnprior)
