/*
 * oracle/trx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, scalar, fp64 CPU restatement of the marginal-likelihood hot path of
 * stevengiacalone/triceratops.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker.
 * The product (triceratops_amd/) never links, imports or falls back to it.
 *
 * What it restates (reference file:line, all under /root/reference/triceratops):
 *   likelihoods.py:302-358   simulate_TP_transit_p   -> trxo_flux_grid(model=TP)
 *   likelihoods.py:361-439   simulate_EB_transit_p   -> trxo_flux_grid(model=EB)
 *   likelihoods.py:443-487   lnL_TP_p                -> trxo_lnl_batch(model=TP)
 *   likelihoods.py:490-539   lnL_EB_p                -> trxo_lnl_batch(model=EB)
 *   likelihoods.py:542-587   lnL_EB_twin_p           -> trxo_lnl_batch(model=EB_TWIN)
 *   likelihoods.py:27-160    scalar simulate_*       -> same with TRXO_FLAG_SCALAR_K
 *   _numerics.py:12-51       _log_mean_exp           -> trxo_log_mean_exp
 *   _numerics.py:54-76       _normalize_probabilities-> trxo_normalize_probabilities
 *   likelihoods.py:24-25,61-71,348-349,414-422: the calls into
 *   pytransit==2.2 QuadraticModel (set_data / evaluate_ps / evaluate_pv) ->
 *   trxo_evaluate_pv.
 *
 * PARITY STATUS
 *   - _log_mean_exp / _normalize_probabilities: PINNED against the imported
 *     reference module and the exact values in the reference's
 *     tests/test_log_mean_exp.py (see tests/test_oracle.py).
 *   - unit conversion, k-tweak, dilution, secondary depth, chi^2/2, +inf rule:
 *     PINNED by golden vectors produced by running the reference's own
 *     likelihoods.py / marginal_likelihoods.py (imported in the build container,
 *     tests/golden/make_golden.py) with this oracle plugged in at the pytransit seam.
 *   - the transit-model arithmetic itself (Kepler orbit + Mandel & Agol 2002
 *     quadratic limb-darkened flux): **PARITY UNPINNED** against pytransit 2.2.
 *     pytransit is a third-party dependency pinned in the reference's setup.py:29
 *     ("pytransit==2.2"); its source is not in /root/reference, it is not installed
 *     and cannot be installed (no network), and no reference test evaluates a light
 *     curve.  This file restates the published algorithm (Mandel & Agol 2002,
 *     ApJ 580, L171, section 4 / Table 1; Kepler's equation) and is pinned instead
 *     against an independent arbitrary-precision quadrature of the limb-darkened
 *     disk (tests/test_oracle.py, mpmath) to <= 1e-13 absolute in flux.
 *
 * Conventions follow the pytransit call sites: pvp columns = [k, t0, p, a, i, e, w],
 * exposure integration = mean of `nsamples` sub-exposures centred at
 * t + exptime*((s-0.5)/nsamples - 0.5), s = 1..nsamples; the far side of the
 * orbit (sin(w+f) < 0) never transits.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TRXO_PI      3.14159265358979323846264338327950288
#define TRXO_TWOPI   6.28318530717958647692528676655900577
#define TRXO_HALFPI  1.57079632679489661923132169163975144

/* cgs constants: astropy.constants (CODATA 2018 / IAU 2015), as used at
 * likelihoods.py:17-21 */
#define TRXO_RSUN    6.957e10
#define TRXO_REARTH  6.3781e8

#define TRXO_MODEL_TP      0
#define TRXO_MODEL_EB      1
#define TRXO_MODEL_EB_TWIN 2

#define TRXO_FLAG_COMPANION_IS_HOST 1
#define TRXO_FLAG_SCALAR_K          2   /* scalar-path k rule, likelihoods.py:121-123 */

/* ------------------------------------------------------------------------- */
/* Bulirsch's general complete elliptic integral
 *   cel(kc,p,a,b) = int_0^{pi/2} (a cos^2 + b sin^2) / ((cos^2 + p sin^2) sqrt(cos^2 + kc^2 sin^2)) dphi
 * for p > 0, kc != 0 (Bulirsch 1969, Numer. Math. 13, 305).  K = cel(kc,1,1,1),
 * E = cel(kc,1,1,kc^2), Pi(n) [1+n sin^2 convention] = cel(kc,1+n,1,1). */
static double trxo_cel(double kc, double p, double a, double b)
{
    const double ca = 1e-8; /* sqrt of the target relative accuracy */
    double e, em, f, g;
    kc = fabs(kc);
    e = kc;
    em = 1.0;
    p = sqrt(p);
    b /= p;
    for (int it = 0; it < 40; ++it) {
        f = a;
        a += b / p;
        g = e / p;
        b += f * g;
        b += b;
        p += g;
        g = em;
        em += kc;
        if (fabs(g - kc) <= g * ca) break;
        kc = 2.0 * sqrt(e);
        e = kc * em;
    }
    return TRXO_HALFPI * (b + a * em) / (em * (em + p));
}

/* Mandel & Agol (2002) quadratic-law flux of a star (radius 1) occulted by a dark
 * disk of radius p at centre-to-centre distance z >= 0.
 *   F = 1 - [(1-c2) le + c2 (ld + 2/3 Theta(p-z)) - c4 ed] / (4 Omega),
 *   c2 = u1 + 2 u2, c4 = -u2, 4 Omega = 1 - u1/3 - u2/6          (their eq. 7-8)
 * le: uniform-source term (their eq. 1); ld, ed: Table 1 / eq. 7.
 * The three contact triangles (sides 1, p, z) are evaluated from the factored
 * side sums f1..f4 so that kappa0, kappa1 and the triangle area are consistent
 * functions of the same rounded inputs near the contact points. */
double trxo_ma_flux(double z, double p, double u1, double u2)
{
    if (isnan(z) || isnan(p)) return NAN;
    if (p <= 0.0) return 1.0;
    const double opp = 1.0 + p;
    if (!(z < opp)) return 1.0;              /* unocculted */
    if (z < 0.0) z = -z;
    if (p >= 1.0 && z <= p - 1.0) return 0.0; /* source fully covered */

    const double z2 = z * z, p2 = p * p;
    const double zmp = z - p;
    const double a = zmp * zmp;
    const double q = (p - z) * (p + z);
    const double omp = 1.0 - p;
    const double f2 = opp - z;        /* 1 + p - z  > 0 */
    const double f3 = 1.0 + zmp;      /* 1 + z - p  > 0 */
    const double f4 = opp + z;        /* 1 + p + z      */
    const double eta2 = 0.5 * p2 * (p2 + 2.0 * z2);
    const double theta = (z < p) ? (2.0 / 3.0) : ((z == p) ? (1.0 / 3.0) : 0.0);
    double le, ld, ed;

    if (p < 1.0 && z <= omp) {
        /* disk entirely inside the stellar limb (Table 1 cases 3, 4, 5, 9, 10) */
        le = p2;
        ed = eta2;
        const double g1 = omp - z;    /* 1 - p - z >= 0 */
        if (g1 == 0.0) {
            /* internal contact: lambda_5 (case 4), Theta terms folded in */
            /* (their -2/3 Theta(p-1/2) and the +2/3 Theta(p-z) of eq. 7 cancel at z = 1-p) */
            ld = (2.0 / (3.0 * TRXO_PI)) * acos(1.0 - 2.0 * p)
               - (4.0 / (9.0 * TRXO_PI)) * (3.0 + 2.0 * p - 8.0 * p2) * sqrt(p * omp);
        } else {
            const double oma = f2 * f3;               /* 1 - a */
            const double kc2 = g1 * f4 / oma;         /* 1 - 4zp/(1-a) */
            const double kc = sqrt(kc2);
            const double al = 1.0 - 5.0 * z2 + p2 + q * q;
            const double be = oma * (z2 + 7.0 * p2 - 4.0);
            double s = trxo_cel(kc, 1.0, al + be, al + be * kc2);
            if (a > 0.0) {
                const double ga = -3.0 * q / a;
                s += trxo_cel(kc, (z + p) * (z + p) / a, ga, ga);
            }
            ld = 2.0 / (9.0 * TRXO_PI * sqrt(oma)) * s + theta;
        }
    } else {
        /* disk crosses the limb (Table 1 cases 2, 6, 7, 8) */
        const double f1 = (p < 1.0) ? (z - omp) : (z + (p - 1.0));   /* z + p - 1 > 0 */
        const double s14 = sqrt(f1 * f4), s23 = sqrt(f2 * f3);
        const double kap0 = 2.0 * atan2(s23, s14);
        const double kap1 = 2.0 * atan2(sqrt(f1 * f2), sqrt(f3 * f4));
        const double area4 = s14 * s23;               /* 4 x triangle area */
        le = (p2 * kap0 + kap1 - 0.5 * area4) / TRXO_PI;
        ed = (kap1 + 2.0 * eta2 * kap0 - 0.25 * (1.0 + 5.0 * p2 + z2) * area4) / TRXO_TWOPI;
        const double fzp = 4.0 * z * p;
        const double kc2 = f1 * f4 / fzp;             /* 1 - (1-a)/(4zp) */
        const double kc = sqrt(kc2);
        const double b = (z + p) * (z + p);
        const double al = (1.0 - b) * (2.0 * b + a - 3.0) - 3.0 * q * (b - 2.0);
        const double be = fzp * (z2 + 7.0 * p2 - 4.0);
        double s = trxo_cel(kc, 1.0, al + be, al + be * kc2);
        if (a > 0.0) {
            const double ga = -3.0 * q / a;
            s += trxo_cel(kc, 1.0 / a, ga, ga);
        }
        ld = s / (9.0 * TRXO_PI * sqrt(p * z)) + theta;
    }
    const double c2 = u1 + 2.0 * u2;
    const double omega4 = 1.0 - u1 / 3.0 - u2 / 6.0;
    return 1.0 - ((1.0 - c2) * le + c2 * ld + u2 * ed) / omega4;
}

/* ------------------------------------------------------------------------- */
/* Kepler's equation E - e sin E = M, solved to fp64 round-off.
 * Newton iteration safeguarded by the bracket [m, m+e] on m = |M| in [0, pi]. */
double trxo_kepler_E(double M, double e)
{
    M = remainder(M, TRXO_TWOPI); /* [-pi, pi] */
    if (e == 0.0) return M;
    const double sgn = (M < 0.0) ? -1.0 : 1.0;
    const double m = fabs(M);
    double lo = m, hi = m + e;
    if (hi > TRXO_PI) hi = TRXO_PI;
    double E = m + e * sin(m) / (1.0 - sin(m + e) + sin(m));
    if (!(E >= lo && E <= hi)) E = 0.5 * (lo + hi);
    for (int it = 0; it < 100; ++it) {
        const double sE = sin(E), cE = cos(E);
        const double f = E - e * sE - m;
        if (f > 0.0) hi = E; else lo = E;
        const double fp = 1.0 - e * cE;
        /* Halley step */
        double dE = -f / (fp + 0.5 * (-f / fp) * e * sE);
        double En = E + dE;
        if (!(En >= lo && En <= hi)) { En = 0.5 * (lo + hi); dE = En - E; }
        E = En;
        if (fabs(dE) <= 2.3e-16 * (1.0 + fabs(E))) break;
    }
    return sgn * E;
}

/* Per-row orbit constants derived once from pvp = [k, t0, p, a, i, e, w]. */
typedef struct {
    double k, t0, n /* 2pi/p */, e, Mtr, ax, ay, bx, by, cosi;
    double wlo, whi;   /* (-inf, inf) in the checker; bench_window_skip.h narrows them */
} trxo_orbit;

#ifdef TRXO_WINDOW_SKIP
#include "bench_window_skip.h"      /* bench.py's cpu_baseline only; not in the checker's default build */
#endif

static void trxo_orbit_init(trxo_orbit* o, double k, double t0, double p, double a,
                            double inc, double e, double w)
{
    const double ftr = TRXO_HALFPI - w;       /* true anomaly at inferior conjunction */
    const double rt = sqrt(1.0 - e * e);
    const double Etr = atan2(rt * sin(ftr), e + cos(ftr));
    o->k = k; o->t0 = t0; o->n = TRXO_TWOPI / p; o->e = e;
    o->Mtr = Etr - e * sin(Etr);
    const double cw = cos(w), sw = sin(w);
    /* X = r cos(w+f) = ax (cosE - e) + bx sinE ;  Y = r sin(w+f) = ay (cosE - e) + by sinE */
    o->ax = a * cw;  o->bx = -a * rt * sw;
    o->ay = a * sw;  o->by =  a * rt * cw;
    o->cosi = cos(inc);
    o->wlo = -INFINITY; o->whi = INFINITY;
#ifdef TRXO_WINDOW_SKIP
    if (trxo_window_skip) trxo_orbit_window(o, Etr, trxo_exptime_hint);
#endif
}

/* projected separation at time t; returns a negative number behind the star */
static double trxo_orbit_z(const trxo_orbit* o, double t)
{
    const double M = o->n * (t - o->t0) + o->Mtr;
    const double E = trxo_kepler_E(M, o->e);
    const double ce = cos(E) - o->e, se = sin(E);
    const double X = o->ax * ce + o->bx * se;
    const double Y = o->ay * ce + o->by * se;
    const double yc = Y * o->cosi;
    const double z = sqrt(X * X + yc * yc);
    return (Y < 0.0) ? -z : z;
}

static double trxo_point_flux(const trxo_orbit* o, double u1, double u2, double t,
                              double exptime, int nsamples)
{
    double acc = 0.0;
#ifdef TRXO_WINDOW_SKIP
    if (trxo_window_skip && trxo_outside_window(o, t)) return 1.0;
#endif
    for (int s = 1; s <= nsamples; ++s) {
        const double off = exptime * (((double)s - 0.5) / (double)nsamples - 0.5);
        const double z = trxo_orbit_z(o, t + off);
        if (z < 0.0 || !(z < 1.0 + o->k)) acc += 1.0;   /* behind / unocculted (NaN -> below) */
        else acc += trxo_ma_flux(z, o->k, u1, u2);
        if (isnan(z)) acc = NAN;
    }
    return acc / (double)nsamples;
}

/* pytransit-shaped: pvp (n,7) row-major [k,t0,p,a,i,e,w], ldc (n,2), out (n,n_time).
 * Stands for QuadraticModel.set_data(time, exptimes, nsamples) + evaluate_pv(pvp, ldc)
 * (likelihoods.py:348-349, 414-415, 421-422). */
void trxo_evaluate_pv(const double* time, int n_time, const double* pvp, const double* ldc,
                      long n, double exptime, int nsamples, double* out)
{
#ifdef TRXO_WINDOW_SKIP
    trxo_exptime_hint = exptime;
#endif
    if (nsamples < 1) nsamples = 1;
#pragma omp parallel for schedule(dynamic, 2)
    for (long r = 0; r < n; ++r) {
        const double* v = pvp + 7 * r;
        trxo_orbit o;
        trxo_orbit_init(&o, v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
        for (int j = 0; j < n_time; ++j)
            out[(size_t)r * n_time + j] =
                trxo_point_flux(&o, ldc[2 * r], ldc[2 * r + 1], time[j], exptime, nsamples);
    }
}

/* ------------------------------------------------------------------------- */
/* One Monte-Carlo row of simulate_TP_transit_p / simulate_EB_transit_p.
 * params for TP  (10): R_p[Rearth], P[d], inc[deg], a[cm], R_s[Rsun], u1, u2, ecc, argp[deg], comp_fr
 *        for EB  (11): R_EB[Rsun], EB_fr, P[d], inc[deg], a[cm], R_s[Rsun], u1, u2, ecc, argp[deg], comp_fr
 * model_out (n_time) may be NULL; returns chi^2/2 (or +inf by the secondary rule). */
typedef struct {
    double k, ksec, P, a_R, inc, e, w, u1, u2;
    double fcomp, feb;
} trxo_row;

static double trxo_k_rule(double k, int scalar_rule)
{
    /* vector path: k[(k - 1.0) < 1e-6] *= 0.999  (likelihoods.py:406, 418: no abs)
     * scalar path: if abs(k - 1.0) < 1e-6: k *= 0.999 (likelihoods.py:122-123) */
    if (scalar_rule) { if (fabs(k - 1.0) < 1e-6) k *= 0.999; }
    else             { if ((k - 1.0) < 1e-6) k *= 0.999; }
    return k;
}

static double trxo_row_eval(int model, int flags, const double* time, const double* flux,
                            int n_time, double sigma, const double* prm, long stride,
                            double exptime, int nsamples, double* model_out,
                            double* secdepth_out)
{
    const int is_host = flags & TRXO_FLAG_COMPANION_IS_HOST;
    const int scalar_k = flags & TRXO_FLAG_SCALAR_K;
    trxo_row r;
    double R_s, argp, comp_fr;
    if (model == TRXO_MODEL_TP) {
        const double R_p = prm[0 * stride];
        r.P = prm[1 * stride]; r.inc = prm[2 * stride];
        const double a = prm[3 * stride];
        R_s = prm[4 * stride]; r.u1 = prm[5 * stride]; r.u2 = prm[6 * stride];
        r.e = prm[7 * stride]; argp = prm[8 * stride]; comp_fr = prm[9 * stride];
        r.k = R_p * TRXO_REARTH / (R_s * TRXO_RSUN);      /* likelihoods.py:340 */
        r.ksec = 0.0; r.feb = 0.0;
        r.a_R = a / (R_s * TRXO_RSUN);                    /* :343 */
    } else {
        const double R_EB = prm[0 * stride], eb_fr = prm[1 * stride];
        r.P = prm[2 * stride]; r.inc = prm[3 * stride];
        const double a = prm[4 * stride];
        R_s = prm[5 * stride]; r.u1 = prm[6 * stride]; r.u2 = prm[7 * stride];
        r.e = prm[8 * stride]; argp = prm[9 * stride]; comp_fr = prm[10 * stride];
        r.feb = eb_fr / (1.0 - eb_fr);                    /* :401 */
        r.k = trxo_k_rule(R_EB / R_s, scalar_k);          /* :405-406 */
        if (scalar_k) r.ksec = 1.0 / r.k;                 /* :137 */
        else r.ksec = trxo_k_rule(R_s / R_EB, 0);         /* :417-418 */
        r.a_R = a / (R_s * TRXO_RSUN);                    /* :409 */
    }
    r.fcomp = comp_fr / (1.0 - comp_fr);                  /* :337, :399 */
    r.inc = r.inc * (TRXO_PI / 180.0);                    /* :344, :410 */
    r.w = (90.0 - argp) * (TRXO_PI / 180.0);              /* :345, :411 */

    trxo_orbit o;
    trxo_orbit_init(&o, r.k, 0.0, r.P, r.a_R, r.inc, r.e, r.w);

    double x_eb = 0.0, f_dil, secdepth = 0.0;
    if (model == TRXO_MODEL_TP) {
        f_dil = is_host ? 1.0 / r.fcomp : r.fcomp;        /* :352-357 */
    } else {
        /* secondary: k -> R_s/R_EB, w + 180 deg, 25 points on [-0.05, 0.05], no supersampling */
        const double wsec = (90.0 - argp + 180.0) * (TRXO_PI / 180.0);  /* :419 */
        trxo_orbit os;
        trxo_orbit_init(&os, r.ksec, 0.0, r.P, r.a_R, r.inc, r.e, wsec);
        double sec = INFINITY;
        int sec_nan = 0;
        for (int j = 0; j < 25; ++j) {
            const double ts = -0.05 + (0.1 / 24.0) * (double)j;
            double tt = ts;
            if (j == 24) tt = 0.05;       /* np.linspace end point is exact */
            const double f = trxo_point_flux(&os, r.u1, r.u2, tt, 0.0, 1);
            if (isnan(f)) sec_nan = 1;
            if (f < sec) sec = f;
        }
        if (sec_nan) sec = NAN;           /* np.min propagates NaN */
        double y_sec;
        if (is_host) {                    /* :427-432 */
            x_eb = r.feb / r.fcomp;
            y_sec = r.fcomp / r.feb;
            f_dil = 1.0 / (r.fcomp + r.feb);
        } else {                          /* :433-438 */
            x_eb = r.feb / 1.0;
            y_sec = 1.0 / r.feb;
            f_dil = r.fcomp / (1.0 + r.feb);
        }
        sec = (sec + y_sec) / (1.0 + y_sec);
        secdepth = 1.0 - (sec + f_dil) / (1.0 + f_dil);
    }
    if (secdepth_out) *secdepth_out = secdepth;

    long double acc = 0.0L;
    const double s2 = sigma * sigma;
    for (int j = 0; j < n_time; ++j) {
        double m = trxo_point_flux(&o, r.u1, r.u2, time[j], exptime, nsamples);
        if (model != TRXO_MODEL_TP) m = (m + x_eb) / (1.0 + x_eb);
        m = (m + f_dil) / (1.0 + f_dil);
        if (model_out) model_out[j] = m;
        if (flux) {
            const double d = flux[j] - m;
            acc += (long double)((d * d) / s2);           /* :486, :537, :586 */
        }
    }
    double h = 0.5 * (double)acc;
    if (model == TRXO_MODEL_EB && !(secdepth < 1.5 * sigma)) h = INFINITY; /* :535-538 */
    return h;
}

/* params: SoA [n_param][n] contiguous (n_param = 10 TP, 11 EB).  out_halfchi2 [n]. */
void trxo_lnl_batch(int model, int flags, const double* time, const double* flux, int n_time,
                    double sigma, const double* params, long n, double exptime, int nsamples,
                    double* out_halfchi2)
{
#ifdef TRXO_WINDOW_SKIP
    trxo_exptime_hint = exptime;
#endif
    if (nsamples < 1) nsamples = 1;
#pragma omp parallel for schedule(dynamic, 2)
    for (long r = 0; r < n; ++r)
        out_halfchi2[r] = trxo_row_eval(model, flags, time, flux, n_time, sigma, params + r, n,
                                        exptime, nsamples, NULL, NULL);
}

/* out_flux [n][n_time]; out_secdepth [n] or NULL (EB only) */
void trxo_flux_grid(int model, int flags, const double* time, int n_time, const double* params,
                    long n, double exptime, int nsamples, double* out_flux, double* out_secdepth)
{
#ifdef TRXO_WINDOW_SKIP
    trxo_exptime_hint = exptime;
#endif
    if (nsamples < 1) nsamples = 1;
#pragma omp parallel for schedule(dynamic, 2)
    for (long r = 0; r < n; ++r) {
        double sd = 0.0;
        (void)trxo_row_eval(model, flags, time, NULL, n_time, 1.0, params + r, n, exptime,
                            nsamples, out_flux + (size_t)r * n_time, &sd);
        if (out_secdepth) out_secdepth[r] = sd;
    }
}

/* the reference's reduction over a materialised grid: 0.5*sum((flux-model)^2/sigma^2, axis=1) */
void trxo_chi2_grid(const double* flux, const double* model, int n_time, long n, double sigma,
                    double* out_halfchi2)
{
    const double s2 = sigma * sigma;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < n; ++r) {
        long double acc = 0.0L;
        for (int j = 0; j < n_time; ++j) {
            const double d = flux[j] - model[(size_t)r * n_time + j];
            acc += (long double)((d * d) / s2);
        }
        out_halfchi2[r] = 0.5 * (double)acc;
    }
}

/* ------------------------------------------------------------------------- */
/* _numerics.py:12-51.  Returns log(mean(exp(logw))) over n_total draws.
 * status: 0 ok, 1 n_total mismatch (the reference raises ValueError). */
int trxo_log_mean_exp(const double* logw, long n, long n_total, double* out)
{
    if (n_total != n) { *out = NAN; return 1; }          /* :40-45 */
    double mx = -INFINITY;
    int any_finite = 0;
    for (long i = 0; i < n; ++i) {
        const double x = logw[i];
        if (isinf(x) && x > 0) { *out = INFINITY; return 0; }   /* :46-47 */
        if (isfinite(x)) { any_finite = 1; if (x > mx) mx = x; }
    }
    if (!any_finite) { *out = -INFINITY; return 0; }     /* :49-50 */
    long double s = 0.0L;
    for (long i = 0; i < n; ++i)
        if (isfinite(logw[i])) s += (long double)exp(logw[i] - mx);
    *out = (double)(logl(s) + (long double)mx - logl((long double)n_total));  /* :51 */
    return 0;
}

/* _numerics.py:54-76.  status: 0 'ok', 1 'all_neginf', 2 'anomaly'. */
int trxo_normalize_probabilities(const double* lnz, int n, double* probs)
{
    int all_neginf = 1, anomaly = 0;
    double mx = -INFINITY;
    for (int i = 0; i < n; ++i) {
        if (isnan(lnz[i]) || (isinf(lnz[i]) && lnz[i] > 0)) anomaly = 1;
        if (!(isinf(lnz[i]) && lnz[i] < 0)) all_neginf = 0;
        if (lnz[i] > mx) mx = lnz[i];
    }
    for (int i = 0; i < n; ++i) probs[i] = 0.0;
    if (anomaly) return 2;
    if (all_neginf) return 1;
    long double s = 0.0L;
    for (int i = 0; i < n; ++i) s += (long double)exp(lnz[i] - mx);
    const double lse = (double)(logl(s) + (long double)mx);
    for (int i = 0; i < n; ++i) probs[i] = exp(lnz[i] - lse);
    return 0;
}

int trxo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void trxo_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
