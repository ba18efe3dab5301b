// Device-side math of the triceratops hot path for gfx950 (wave64, fp64 VALU).
//
// Everything here is scalar-per-lane fp64: the path is elementwise + reduction (no MFMA
// shape anywhere), bounded by fp64 VALU issue and the div/sqrt/transcendental sequences,
// not by HBM.  Design notes (DESIGN.md has the full derivations):
//   * Kepler's equation is solved once per light-curve point; the S sub-exposures of that
//     point are reached by Newton steps on dE using Taylor kernels for sin(dE), cos(dE)-1
//     (|dE| ~ 1e-4 rad), so no trig range reduction runs inside the supersample loop.
//   * the two Bulirsch `cel` integrals of a Mandel-Agol evaluation share one AGM loop.
//   * a per-row mean-anomaly window (analytic bound of |X| < 1+k around inferior
//     conjunction) lets out-of-transit points return exactly 1.0 without touching the orbit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace trx {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kTwoPi = 6.28318530717958647692528676655900577;
constexpr double kHalfPi = 1.57079632679489661923132169163975144;
// cgs constants of astropy.constants as used at likelihoods.py:17-21
constexpr double kRsun = 6.957e10;
constexpr double kRearth = 6.3781e8;

constexpr int kSecPoints = 25;  // likelihoods.py:135, 421: np.linspace(-0.05, 0.05, 25)

// ---------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// cel(kc,1,a1,b1) + cel(kc,p2,g,g) (Bulirsch 1969) with the kc/em recurrence shared.
__device__ __forceinline__ double cel_pair(double kc, double a1, double b1, double p2, double g2)
{
    double e = kc, em = 1.0, q = kc;
    double p1 = 1.0;
    double pp = sqrt(p2);
    double a2 = g2, b2 = g2 / pp;
#pragma unroll 1
    for (int it = 0; it < 40; ++it) {
        const double r1 = 1.0 / p1, r2 = 1.0 / pp;
        double f = a1, g = e * r1;
        a1 = fma(b1, r1, a1);
        b1 = fma(f, g, b1);
        b1 += b1;
        p1 += g;
        f = a2;
        g = e * r2;
        a2 = fma(b2, r2, a2);
        b2 = fma(f, g, b2);
        b2 += b2;
        pp += g;
        g = em;
        em += q;
        if (fabs(g - q) <= g * 1e-8) break;
        q = 2.0 * sqrt(e);
        e = q * em;
    }
    return kHalfPi * ((b1 + a1 * em) / (em * (em + p1)) + (b2 + a2 * em) / (em * (em + pp)));
}

// Limb-darkening weights of one row: F = 1 - (cle*le + cld*ld + ced*ed)
struct Limb {
    double cle, cld, ced;
};

__device__ __forceinline__ Limb limb_weights(double u1, double u2)
{
    const double c2 = u1 + 2.0 * u2;
    const double om4 = 1.0 - u1 / 3.0 - u2 / 6.0;
    Limb L;
    L.cle = (1.0 - c2) / om4;
    L.cld = c2 / om4;
    L.ced = u2 / om4;
    return L;
}

// Mandel & Agol (2002) quadratic-law flux for 0 <= z < 1+p, p > 0 (callers handle the
// unocculted side).  Case analysis and factored contact-triangle form: DESIGN.md section 4.
__device__ double ma_flux(double z, double p, const Limb& L)
{
    if (p >= 1.0 && z <= p - 1.0) return 0.0;
    const double z2 = z * z, p2 = p * p;
    const double zmp = z - p;
    const double a = zmp * zmp;
    const double q = (p - z) * (p + z);
    const double opp = 1.0 + p, omp = 1.0 - p;
    const double f2 = opp - z, f3 = 1.0 + zmp, f4 = opp + z;
    const double eta2 = 0.5 * p2 * (p2 + 2.0 * z2);
    const double theta = (z < p) ? (2.0 / 3.0) : ((z == p) ? (1.0 / 3.0) : 0.0);
    const double t7 = z2 + 7.0 * p2 - 4.0;
    // the term -3 q / a * Pi(...) vanishes in the limit z -> p (a == 0): drop it there
    const double ga = (a > 0.0) ? (-3.0 * q / a) : 0.0;
    double le, ld, ed;
    if (p < 1.0 && z <= omp) {
        le = p2;
        ed = eta2;
        const double g1 = omp - z;
        if (g1 == 0.0) {
            ld = (2.0 / (3.0 * kPi)) * acos(1.0 - 2.0 * p)
               - (4.0 / (9.0 * kPi)) * (3.0 + 2.0 * p - 8.0 * p2) * sqrt(p * omp);
        } else {
            const double oma = f2 * f3;
            const double kc2 = g1 * f4 / oma;
            const double al = 1.0 - 5.0 * z2 + p2 + q * q;
            const double be = oma * t7;
            const double P = (a > 0.0) ? ((z + p) * (z + p) / a) : 1.0;
            const double s = cel_pair(sqrt(kc2), al + be, fma(be, kc2, al), P, ga);
            ld = 2.0 / (9.0 * kPi * sqrt(oma)) * s + theta;
        }
    } else {
        const double f1 = (p < 1.0) ? (z - omp) : (z + (p - 1.0));
        const double s14 = sqrt(f1 * f4), s23 = sqrt(f2 * f3);
        const double kap0 = 2.0 * atan2(s23, s14);
        const double kap1 = 2.0 * atan2(sqrt(f1 * f2), sqrt(f3 * f4));
        const double area4 = s14 * s23;
        le = (p2 * kap0 + kap1 - 0.5 * area4) * (1.0 / kPi);
        ed = (kap1 + 2.0 * eta2 * kap0 - 0.25 * (1.0 + 5.0 * p2 + z2) * area4) * (1.0 / kTwoPi);
        const double fzp = 4.0 * z * p;
        const double kc2 = f1 * f4 / fzp;
        const double b = (z + p) * (z + p);
        const double al = (1.0 - b) * (2.0 * b + a - 3.0) - 3.0 * q * (b - 2.0);
        const double be = fzp * t7;
        const double P = (a > 0.0) ? (1.0 / a) : 1.0;
        const double s = cel_pair(sqrt(kc2), al + be, fma(be, kc2, al), P, ga);
        ld = s / (9.0 * kPi * sqrt(p * z)) + theta;
    }
    return 1.0 - (L.cle * le + L.cld * ld + L.ced * ed);
}

// ---------------------------------------------------------------------------------------
// Kepler's equation, full solve: bracket-safeguarded Halley on m = |M| in [0, pi].
// Returns E and (sin E, cos E).
__device__ __forceinline__ void kepler_full(double M, double e, double& E, double& sE, double& cE)
{
    M = remainder(M, kTwoPi);
    const double sgn = (M < 0.0) ? -1.0 : 1.0;
    const double m = fabs(M);
    double lo = m, hi = fmin(m + e, kPi);
    double sm, cm;
    sincos(m, &sm, &cm);
    double x = m + e * sm * rsqrt(fma(e, e - 2.0 * cm, 1.0));
    if (!(x >= lo && x <= hi)) x = 0.5 * (lo + hi);
    double s = sm, c = cm;
    bool done = (e == 0.0);
    if (done) x = m;
#pragma unroll 1
    for (int it = 0; it < 24; ++it) {
        if (__all(done)) break;
        sincos(x, &s, &c);
        const double f = x - e * s - m;
        if (f > 0.0) hi = x; else lo = x;
        const double fp = 1.0 - e * c;
        double dx = -f / (fp + 0.5 * (-f / fp) * e * s);
        double xn = x + dx;
        if (!(xn >= lo && xn <= hi)) { xn = 0.5 * (lo + hi); dx = xn - x; }
        if (!done) x = xn;
        done = done || (fabs(dx) <= 2.3e-16 * (1.0 + fabs(x))) || !(dx == dx);
    }
    sincos(x, &s, &c);
    E = sgn * x;
    sE = sgn * s;
    cE = c;
}

// Advance the solution (E, sinE, cosE) of E - e sinE = M to M + dM for small dM:
// Newton on g(d) = d - e (sinE (cos d - 1) + cosE sin d) - dM with Taylor kernels.
// Returns false when |d| is too large for the series (caller falls back to kepler_full).
__device__ __forceinline__ bool kepler_step(double dM, double e, double& E, double& sE, double& cE)
{
    const double rho = 1.0 / (1.0 - e * cE);
    double d = dM * rho;
    if (!(fabs(d) < 0.08)) return false;
    double ds = 0.0, dc = 0.0;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const double d2 = d * d;
        const double sd = d * fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 362880.0, -1.0 / 5040.0),
                                                    1.0 / 120.0), -1.0 / 6.0), 1.0);
        const double c1 = d2 * fma(d2, fma(d2, fma(d2, fma(d2, -1.0 / 3628800.0, 1.0 / 40320.0),
                                                     -1.0 / 720.0), 1.0 / 24.0), -0.5);
        ds = fma(sE, c1, cE * sd);   // sin(E+d) - sin E
        dc = fma(cE, c1, -sE * sd);  // cos(E+d) - cos E
        const double g = d - e * ds - dM;
        const double gp = 1.0 - e * (cE + dc);
        const double step = g / gp;
        d -= step;
        if (__all(fabs(step) <= 1e-17 + 1.2e-16 * fabs(d))) break;
    }
    E += d;
    sE += ds;
    cE += dc;
    return true;
}

// ---------------------------------------------------------------------------------------
// Per-row constants staged in LDS (one row = one Monte-Carlo draw).
struct RowC {
    double k, t0, nmot, e, Mtr;
    double ax, ay, bx, by, cosi;
    double wlo, whi;          // mean-anomaly window (relative to conjunction), margins included
    double cle, cld, ced;     // limb-darkening weights
    double xeb, fdil;         // dilution: m = (m + xeb)/(1 + xeb) [EB]; m = (m + fdil)/(1 + fdil)
    double excl;              // 1.0 when the EB secondary rule excludes the draw (+inf), else 0
};
constexpr int kRowDoubles = sizeof(RowC) / sizeof(double);

// Orbit constants from pytransit-shaped (k, t0, p, a, i, e, w).
__device__ __forceinline__ void orbit_init(RowC& c, double k, double t0, double per, double a,
                                           double inc, double e, double w, double exptime)
{
    const double ftr = kHalfPi - w;
    const double rt = sqrt(1.0 - e * e);
    double sf, cf;
    sincos(ftr, &sf, &cf);
    const double Etr = atan2(rt * sf, e + cf);
    double sEt, cEt;
    sincos(Etr, &sEt, &cEt);
    double sw, cw;
    sincos(w, &sw, &cw);
    c.k = k;
    c.t0 = t0;
    c.nmot = kTwoPi / per;
    c.e = e;
    c.Mtr = Etr - e * sEt;
    c.ax = a * cw;
    c.bx = -a * rt * sw;
    c.ay = a * sw;
    c.by = a * rt * cw;
    c.cosi = cos(inc);
    // Window: X(E) = ax (cosE - e) + bx sinE = A cos(E - phi) - ax e; transit needs |X| < 1+k.
    const double R = (1.0 + k) * (1.0 + 1e-9) + 1e-12;
    const double A = sqrt(c.ax * c.ax + c.bx * c.bx);
    const double phi = atan2(c.bx, c.ax);
    const double clo = (c.ax * e - R) / A, chi = (c.ax * e + R) / A;
    const double psi = remainder(Etr - phi, kTwoPi);
    double plo, phi2;  // arc [plo, phi2] of psi = E - phi containing psi
    const bool open_hi = !(chi < 1.0), open_lo = !(clo > -1.0);
    if (open_hi && open_lo) {
        plo = psi - kTwoPi;
        phi2 = psi + kTwoPi;
    } else if (open_hi) {
        const double x = acos(clo);
        plo = -x;
        phi2 = x;
    } else if (open_lo) {
        const double x = acos(chi);
        if (psi >= 0.0) { plo = x; phi2 = kTwoPi - x; } else { plo = x - kTwoPi; phi2 = -x; }
    } else {
        const double x0 = acos(chi), x1 = acos(clo);
        if (psi >= 0.0) { plo = x0; phi2 = x1; } else { plo = -x1; phi2 = -x0; }
    }
    const double Elo = Etr + (plo - psi), Ehi = Etr + (phi2 - psi);
    const double mg = 0.5 * fabs(c.nmot * exptime) * (1.0 + 1e-9) + 1e-11;
    c.wlo = (Elo - Etr) - e * (sin(Elo) - sEt) - mg;
    c.whi = (Ehi - Etr) - e * (sin(Ehi) - sEt) + mg;
    // NaN anywhere (invalid draw) => comparisons below are false => the point is evaluated
    // and the NaN propagates to the result exactly as in the plain restatement.
}

__device__ __forceinline__ bool in_window(const RowC& c, double dMc)
{
    // dMc in [-pi, pi]: mean anomaly of the exposure centre relative to conjunction
    const bool out = ((dMc < c.wlo) && !(dMc + kTwoPi <= c.whi)) ||
                     ((dMc > c.whi) && !(dMc - kTwoPi >= c.wlo));
    return !out;
}

// Mean model flux of one exposure (centre t), S sub-exposures: the body of
// pytransit's evaluate_pv for one (row, time) cell.
template <bool STEP>
__device__ __forceinline__ double exposure_flux(const RowC& c, const Limb& L, double t,
                                                double exptime, int S)
{
    const double tc = t - c.t0;
    const double dMc = remainder(c.nmot * tc, kTwoPi);
    if (!in_window(c, dMc)) return 1.0;
    const double opp = 1.0 + c.k;
    const double opp2 = opp * opp;
    const double dS = (double)S;
    double acc = 0.0;
    double E = 0.0, sE = 0.0, cE = 1.0, Mprev = 0.0;
#pragma unroll 1
    for (int s = 1; s <= S; ++s) {
        const double off = exptime * (((double)s - 0.5) / dS - 0.5);
        const double M = c.nmot * ((t + off) - c.t0) + c.Mtr;
        bool stepped = false;
        if (STEP && s > 1) stepped = kepler_step(M - Mprev, c.e, E, sE, cE);
        if (!stepped) kepler_full(M, c.e, E, sE, cE);
        Mprev = M;
        const double ce = cE - c.e;
        const double X = fma(c.ax, ce, c.bx * sE);
        const double Y = fma(c.ay, ce, c.by * sE);
        const double yc = Y * c.cosi;
        const double z2 = fma(X, X, yc * yc);
        double f = 1.0;
        if (Y >= 0.0 && z2 < opp2) f = ma_flux(sqrt(z2), c.k, L);
        else if (z2 != z2) f = z2;
        acc += f;
    }
    return acc / dS;
}

}  // namespace trx
