// Device-side math of the triceratops hot path for gfx950 (wave64, fp64 VALU).
//
// Everything here is scalar-per-lane fp64: the path is elementwise + reduction (no MFMA
// shape anywhere), bounded by fp64 VALU issue and the div/sqrt/transcendental sequences,
// not by HBM.  Design notes (DESIGN.md has the full derivations):
//   * Kepler's equation is solved once per light-curve point (at the exposure centre); the nodes
//     or sub-exposures of that point are reached by Newton steps on dE using Taylor kernels for
//     sin(dE), cos(dE)-1 (|dE| ~ 1e-3 rad), so no trig range reduction runs in the node loop.
//   * the S-point exposure average is taken from the 3-9 point Gauss rule of that measure wherever the model is
//     analytic over the exposure (TierTable / plan_cell), from all S sub-exposures near the limb
//     contacts, and is exactly 1 off the disc.
//   * the two Bulirsch `cel` integrals of a Mandel-Agol evaluation share one AGM loop and
//     one reciprocal per iteration.
//   * a per-row mean-anomaly window (analytic bound of |X| < 1+k around inferior
//     conjunction) lets out-of-transit points return exactly 1.0 without touching the orbit.
//   * reciprocal / square root / sincos are short fp64 sequences seeded by v_rcp_f64 /
//     v_rsq_f64 (about 1 ulp, not correctly rounded): the IEEE div/sqrt expansions and the
//     Payne-Hanek path of the library sincos cost registers and issue slots this kernel needs.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace trx {

// Trip census (build with -DTRX_CENSUS; profiles/r05/isa_histogram.py weights the static opcode counts of cells_kernel's
// regions with it).  Not compiled into the product library.  A count is added once per WAVE execution of the place it
// stands at (the first active lane adds).
#ifdef TRX_CENSUS
__device__ unsigned long long g_census[32];
__device__ __forceinline__ void census_add(int slot, unsigned long long n = 1ull)
{
    const unsigned long long m = __ballot(1);
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == __ffsll((long long)m) - 1) atomicAdd(&g_census[slot], n);
}
#define TRX_CENSUS_ADD(slot, n) census_add(slot, n)
#else
#define TRX_CENSUS_ADD(slot, n)
#endif
// census slots
constexpr int kCenBatch = 0, kCenWindowTrip = 1, kCenChunk0 = 2, kCenChunk1 = 3, kCenPass = 4, kCenPairTrip = 5,
              kCenPairLanes = 6, kCenFluxTrip = 7, kCenAgmTrip = 8, kCenKeplerFullPair = 9, kCenKeplerFullPlan = 10,
              kCenFluxLanes = 11, kCenContactTrip = 12, kCenCrossingTrip = 13, kCenInsideTrip = 14;

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
// short fp64 sequences (finite, normal-range arguments; a few ulp).
// Measured on gfx950 (profiles/micro/): v_rcp_f64 / v_rsq_f64 cost 16 cycles per wave, an fp64
// fma/mul/add 4, and the seeds are good to 2^-24.2.  One third-order step (e + e^2, resp.
// r + 1.5 r^2) therefore reaches 2^-72 in one op less than two Newton steps.
__device__ __forceinline__ double rcp_fast(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// one Newton step only (2^-48): for callers that iterate on the result anyway
__device__ __forceinline__ double rcp_nr1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

// sqrt(x) and 1/sqrt(x) for x > 0 (Goldschmidt from v_rsq_f64, third order)
__device__ __forceinline__ void sqrt_rsqrt_pos(double x, double& s, double& rs)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-g, h, 0.5);
    const double t = fma(1.5 * r, r, r);
    s = fma(g, t, g);
    rs = 2.0 * fma(h, t, h);
}

__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-g, h, 0.5);
    return fma(g, fma(1.5 * r, r, r), g);
}

// the same with x == 0 allowed: (0, inf)
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs)
{
    double g, h;
    sqrt_rsqrt_pos(x, g, h);
    s = (x == 0.0) ? 0.0 : g;
    rs = (x == 0.0) ? INFINITY : h;
}

__device__ __forceinline__ double sqrt_fast(double x)
{
    const double g = sqrt_pos(x);
    return (x == 0.0) ? 0.0 : g;
}

// x reduced to [-pi, pi] (two-word 2 pi under fma; error ~1e-16 |x| for |x| < ~1e9)
// x*y + C for a literal C, with C held in a scalar register pair.  Left to itself the compiler
// forms v_fmac_f64 for a Horner step and first copies C into the accumulator VGPR pair: two
// extra VALU issues per step on a VALU-issue-bound kernel.  The s_mov pair is free (scalar unit).
__device__ __forceinline__ double fma_k(double x, double y, double C)
{
#ifdef TRX_NO_SGPR_CONSTS
    return fma(x, y, C);
#else
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(C));
    return r;
#endif
}

__device__ __forceinline__ double reduce_2pi(double x)
{
    const double n = rint(x * 0.15915494309189533577);   // 1/(2 pi)
    return fma(-n, 2.44929359829470641435e-16, fma(-n, 6.28318530717958623200e+00, x));
}

// sin and cos for |x| up to ~1e5 (orbit angles): Cody-Waite reduction by pi/2 with a
// two-word pi/2 under fma, fdlibm kernel polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_red(double x, double& s, double& c)
{
    const double n = rint(x * 0.63661977236758134308);   // 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, 1.58969099521155010221e-10,
                      -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                      -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                      -1.66666666666666324348e-01);
    const double pc = fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, -1.13596475577881948265e-11,
                      2.08757232129817482790e-09), -2.75573143513906633035e-07),
                      2.48015872894767294178e-05), -1.38888888888741095749e-03),
                      4.16666666666666019037e-02);
    const double sr = fma(z * r, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)n & 3;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// atan(x) for x >= 0 (x = +inf allowed): the four-breakpoint argument reduction and the
// degree-10 odd minimax polynomial of fdlibm's s_atan.c, selected without branches; one
// reciprocal.
__device__ __forceinline__ double atan_pos(double x)
{
    const bool r0 = x < 0.4375, r1 = x < 0.6875, r2 = x < 1.1875, r3 = x < 2.4375;
    // t = num / den
    const double num = r0 ? x : (r1 ? fma(2.0, x, -1.0) : (r2 ? x - 1.0 : (r3 ? x - 1.5 : -1.0)));
    const double den = r0 ? 1.0 : (r1 ? 2.0 + x : (r2 ? x + 1.0 : (r3 ? fma(1.5, x, 1.0) : x)));
    const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : (r2 ? 7.85398163397448278999e-01
                         : (r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00)));
    const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : (r2 ? 3.06161699786838301793e-17
                         : (r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17)));
    const double t = (den > 1.7e308) ? 0.0 : num * rcp_fast(den);
    const double z = t * t, w = z * z;
    const double s1 = z * fma_k(w, fma_k(w, fma_k(w, fma_k(w, fma_k(w, 1.62858201153657823623e-02,
                      4.97687799461593236017e-02), 6.66107313738753120669e-02),
                      9.09088713343650656196e-02), 1.42857142725034663711e-01),
                      3.33333333333329318027e-01);
    const double s2 = w * fma_k(w, fma_k(w, fma_k(w, fma_k(w, -3.65315727442169155270e-02,
                      -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                      -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    return hi - ((t * (s1 + s2) - lo) - t);
}

// The same function with the argument reduction's constants taken from a 5 x 6 table (in LDS: cells_kernel) by
// range index instead of four nested selects per constant: t = (alpha x + beta) / (gamma x + delta), then
// hi, lo.  The same operations on the same numbers as atan_pos -- bit for bit -- in ~25 instructions less.
constexpr int kAtanRanges = 5, kAtanCols = 6;
// (alpha, beta, gamma, delta, hi, lo) of the ranges x < 0.4375, < 0.6875, < 1.1875, < 2.4375 and the rest
__device__ const double kAtanTable[kAtanRanges * kAtanCols] = {
    1.0,  0.0, 0.0, 1.0, 0.0,                        0.0,
    2.0, -1.0, 1.0, 2.0, 4.63647609000806093515e-01, 2.26987774529616870924e-17,
    1.0, -1.0, 1.0, 1.0, 7.85398163397448278999e-01, 3.06161699786838301793e-17,
    1.0, -1.5, 1.5, 1.0, 9.82793723247329054082e-01, 1.39033110312309984516e-17,
    0.0, -1.0, 1.0, 0.0, 1.57079632679489655800e+00, 6.12323399573676603587e-17};

__device__ __forceinline__ double atan_pos_tab(double x, const double* tab)
{
    const int i = (x >= 0.4375 ? 1 : 0) + (x >= 0.6875 ? 1 : 0) + (x >= 1.1875 ? 1 : 0) + (x >= 2.4375 ? 1 : 0);
    const double* r = tab + kAtanCols * i;       // (NaN: range 0, as in atan_pos where every comparison is false... see below)
    const double num = fma(r[0], x, r[1]);
    const double den = fma(r[2], x, r[3]);
    const double t = (den > 1.7e308) ? 0.0 : num * rcp_fast(den);
#ifdef TRX_ATAN_LATE_LOADS
    // (hi, lo) are needed last: tied to t, their loads cannot be hoisted above the division and do not hold four
    // registers across it and the polynomial (the batched bounded instantiation sits exactly at its 96-register limit)
    int i_late = i;
    asm("" : "+v"(i_late) : "v"(t));
    const double* r_late = tab + kAtanCols * i_late;
#else
    const double* r_late = r;
#endif
    const double z = t * t, w = z * z;
    const double s1 = z * fma_k(w, fma_k(w, fma_k(w, fma_k(w, fma_k(w, 1.62858201153657823623e-02,
                      4.97687799461593236017e-02), 6.66107313738753120669e-02),
                      9.09088713343650656196e-02), 1.42857142725034663711e-01),
                      3.33333333333329318027e-01);
    const double s2 = w * fma_k(w, fma_k(w, fma_k(w, fma_k(w, -3.65315727442169155270e-02,
                      -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                      -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    return r_late[4] - ((t * (s1 + s2) - r_late[5]) - t);
}

// ---------------------------------------------------------------------------------------
// cel(kc,1,a1,b1) + cel(kc,p2,g2,g2) (Bulirsch 1969): the kc/em recurrence is shared and the
// two p-sequences use one reciprocal per iteration.  Returns the sum of the two integrals.
// The second integral is passed pre-scaled: pp = sqrt(p2), a2 = g2, b2 = g2 / sqrt(p2).
__device__ __forceinline__ double cel_pair(double kc, double a1, double b1, double pp, double a2,
                                           double b2)
{
    double e = kc, em = 1.0, q = kc;
    double p1 = 1.0;
    // one Bulirsch step; true when the kc/em recurrence has converged
    auto step = [&]() -> bool {
        const double r = rcp_fast(p1 * pp);
        const double r1 = r * pp, r2 = r * p1;
        const double g1 = e * r1, g2 = e * r2;
        const double t1 = fma(a1, g1, b1), t2 = fma(a2, g2, b2);
        a1 = fma(b1, r1, a1);
        a2 = fma(b2, r2, a2);
        b1 = t1 + t1;
        b2 = t2 + t2;
        p1 += g1;
        pp += g2;
        const double g = em;
        em += q;
        // the gap squares every step: stopping at a relative gap of 5e-7 leaves ~3e-14 relative
        // on the integrals (Bulirsch's sqrt(eps) = 1e-8 costs 2.4 % more time for nothing; 2e-6
        // was 0.8 % faster but let grazing small-planet rows drift by 3e-13)
        if (fabs(g - q) <= g * 5e-7) return true;
        q = 2.0 * sqrt_pos(e);
        e = q * em;
        return false;
    };
    // two steps per trip: the a/b pairs swap registers every step, so a single-step loop pays
    // four v_mov_b64 per iteration (4 cycles each, as much as an fp64 fma) to rotate them back
#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
        TRX_CENSUS_ADD(kCenAgmTrip, 1);
        if (step()) break;
        if (step()) break;
    }
    const double d1 = em * (em + p1), d2 = em * (em + pp);
    const double num = fma(fma(a1, em, b1), d2, fma(a2, em, b2) * d1);
    return kHalfPi * num * rcp_fast(d1 * d2);
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
// Both regions (disk inside the limb / crossing it) feed ONE cel_pair call.
// TAB: the arctangent's range constants come from the 5 x 6 table at `atab` (in LDS: cells_kernel).  A template
// parameter, not a test of the pointer: the compiler cannot tell that an LDS address is not null and kept BOTH
// arctangents behind a branch -- 100 instructions of every instantiation that nobody executed.
template <int TAB = 0>
__device__ __forceinline__ double ma_flux(double z, double p, const Limb& L, const double* atab = nullptr)
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
    const bool inside = (p < 1.0 && z <= omp);
    // second integral: gamma * Pi(n) with gamma = -3 q / a, a = (z-p)^2.  With r = 1/|z-p|:
    // sqrt(1+n) and gamma/sqrt(1+n) have closed forms (no sqrt, one reciprocal); the term
    // vanishes in the limit z -> p and is dropped there.
    const double zpp = z + p;
    const double azmp = fabs(zmp);
    const bool nz = azmp > 0.0;
    const double r = nz ? rcp_fast(azmp) : 0.0;
    const double m3 = nz ? ((p > z) ? -3.0 : 3.0) : 0.0;      // -3 sign(p - z)
    const double b = zpp * zpp;
    double le, ed, kc2, al, be, pp, a2, b2, scale;
    bool contact = false;
    TRX_CENSUS_ADD(kCenFluxTrip, 1);
    TRX_CENSUS_ADD(kCenFluxLanes, (unsigned long long)__popcll(__ballot(1)));
    if (inside) {
        TRX_CENSUS_ADD(kCenInsideTrip, 1);
        le = p2;
        ed = eta2;
        const double g1 = omp - z;
        contact = (g1 == 0.0);
        const double oma = f2 * f3;
        double so, rso;
        sqrt_rsqrt_pos(oma, so, rso);
        kc2 = g1 * f4 * rso * rso;
        al = 1.0 - 5.0 * z2 + p2 + q * q;
        be = oma * t7;
        pp = nz ? zpp * r : 1.0;                 // sqrt(b / a)
        a2 = m3 * pp;                             // -3 q / a
        b2 = m3;                                  // a2 / pp
        scale = (2.0 / (9.0 * kPi)) * rso;
    } else {
        TRX_CENSUS_ADD(kCenCrossingTrip, 1);
        const double f1 = (p < 1.0) ? (z - omp) : (z + (p - 1.0));
        // contact triangle (sides 1, p, z): one square root, area4 = 4 x area; the half-angle
        // tangents are sqrt(f2 f3 / (f1 f4)) = area4/(f1 f4) and sqrt(f1 f2 / (f3 f4)) = area4/(f3 f4)
        const double f14 = f1 * f4, f34 = f3 * f4;
        const double area4 = sqrt_fast(f14 * (f2 * f3));
        const double x0 = (f14 > 0.0) ? area4 * rcp_fast(f14) : INFINITY, x1 = area4 * rcp_fast(f34);
        // (TAB 2: decided by a test of the pointer at run time -- both versions in the binary; see trx_cells.hpp TRX_ATAN_TAB)
        const bool tab = TAB == 1 || (TAB == 2 && atab != nullptr);
        const double kap0 = 2.0 * (tab ? atan_pos_tab(x0, atab) : atan_pos(x0));
        const double kap1 = 2.0 * (tab ? atan_pos_tab(x1, atab) : atan_pos(x1));
        le = (p2 * kap0 + kap1 - 0.5 * area4) * (1.0 / kPi);
        ed = (kap1 + 2.0 * eta2 * kap0 - 0.25 * (1.0 + 5.0 * p2 + z2) * area4) * (1.0 / kTwoPi);
        const double fzp = 4.0 * z * p;
        double sz, rsz;
        sqrt_rsqrt_pos(fzp, sz, rsz);
        kc2 = f1 * f4 * rsz * rsz;
        al = (1.0 - b) * (2.0 * b + a - 3.0) - 3.0 * q * (b - 2.0);
        be = fzp * t7;
        pp = nz ? r : 1.0;                        // sqrt(1 / a)
        b2 = m3 * zpp;                            // a2 / pp
        a2 = b2 * r;                              // -3 q / a
        scale = (2.0 / (9.0 * kPi)) * rsz;   // 1/(9 pi sqrt(p z))
    }
    double ld;
    if (contact) {
        ld = (2.0 / (3.0 * kPi)) * acos(1.0 - 2.0 * p)
           - (4.0 / (9.0 * kPi)) * (3.0 + 2.0 * p - 8.0 * p2) * sqrt(p * omp);
    } else {
        const double s = cel_pair(sqrt_fast(kc2), al + be, fma(be, kc2, al), pp, a2, b2);
        ld = fma(scale, s, theta);
    }
    return 1.0 - (L.cle * le + L.cld * ld + L.ced * ed);
}

// ---------------------------------------------------------------------------------------
// Mixed-precision variant (BASELINE config 5): the orbit and the geometric differences
// (z-p, 1-p-z, 1+p-z ...) stay fp64 -- they carry the cancellations -- and everything after
// them (products, square roots, the cel loop, atan, the case combination) runs in fp32 on the
// full-rate hardware reciprocal / rsqrt / sqrt.  The result is a flux deficit of relative
// accuracy ~1e-6, i.e. ~1e-7 absolute in flux for the reference's depths.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsqf_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrtf_(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float atan_pos_f(float x)
{
    const bool r0 = x < 0.4375f, r1 = x < 0.6875f, r2 = x < 1.1875f, r3 = x < 2.4375f;
    const float num = r0 ? x : (r1 ? fmaf(2.0f, x, -1.0f) : (r2 ? x - 1.0f : (r3 ? x - 1.5f : -1.0f)));
    const float den = r0 ? 1.0f : (r1 ? 2.0f + x : (r2 ? x + 1.0f : (r3 ? fmaf(1.5f, x, 1.0f) : x)));
    const float hi = r0 ? 0.0f : (r1 ? 4.6364760900e-01f : (r2 ? 7.8539816340e-01f
                        : (r3 ? 9.8279372325e-01f : 1.5707963268e+00f)));
    const float t = (den > 3.0e38f) ? 0.0f : num * rcpf_(den);
    const float z = t * t;
    // atan(t) = t + t z q(z), z = t^2 <= 0.4375^2: degree-4 Chebyshev fit of q, |error| < 2e-9
    const float q = fmaf(z, fmaf(z, fmaf(z, fmaf(z, -0.062244711499f, 0.10671716232f),
                                         -0.14256975819f), 0.19999330138f), -0.33333330811f);
    return hi + fmaf(t * z, q, t);
}

__device__ __forceinline__ float cel_pair_f(float kc, float a1, float b1, float pp, float a2, float b2)
{
    float e = kc, em = 1.0f, q = kc, p1 = 1.0f;
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
        const float r = rcpf_(p1 * pp);
        const float r1 = r * pp, r2 = r * p1;
        float f = a1, g = e * r1;
        a1 = fmaf(b1, r1, a1);
        b1 = fmaf(f, g, b1);
        b1 += b1;
        p1 += g;
        f = a2;
        g = e * r2;
        a2 = fmaf(b2, r2, a2);
        b2 = fmaf(f, g, b2);
        b2 += b2;
        pp += g;
        g = em;
        em += q;
        if (fabsf(g - q) <= g * 3.5e-4f) break;
        q = 2.0f * sqrtf_(e);
        e = q * em;
    }
    const float d1 = em * (em + p1), d2 = em * (em + pp);
    return 1.57079632679f * fmaf(fmaf(a1, em, b1), d2, fmaf(a2, em, b2) * d1) * rcpf_(d1 * d2);
}

__device__ __forceinline__ double ma_flux_f32(double z, double p, const Limb& L)
{
    if (p >= 1.0 && z <= p - 1.0) return 0.0;
    // fp64: the differences
    const double omp = 1.0 - p, opp = 1.0 + p, zmp_d = z - p;
    const bool inside = (p < 1.0 && z <= omp);
    const double d1_d = inside ? (omp - z) : ((p < 1.0) ? (z - omp) : (z + (p - 1.0)));   // g1 or f1
    if (inside && d1_d == 0.0) {
        const double p2 = p * p;
        const double ld = (2.0 / (3.0 * kPi)) * acos(1.0 - 2.0 * p)
                        - (4.0 / (9.0 * kPi)) * (3.0 + 2.0 * p - 8.0 * p2) * sqrt(p * omp);
        return 1.0 - (L.cle * p2 + L.cld * ld + L.ced * (0.5 * p2 * (p2 + 2.0 * z * z)));
    }
    // fp32 from here
    const float zf = (float)z, pf = (float)p, zmp = (float)zmp_d, d1 = (float)d1_d;
    const float f2 = (float)(opp - z), f3 = (float)(1.0 + zmp_d), f4 = (float)(opp + z);
    const float z2 = zf * zf, p2 = pf * pf, a = zmp * zmp, zpp = zf + pf;
    const float q = -zmp * zpp;
    const float eta2 = 0.5f * p2 * (p2 + 2.0f * z2);
    const float theta = (zmp_d < 0.0) ? (2.0f / 3.0f) : ((zmp_d == 0.0) ? (1.0f / 3.0f) : 0.0f);
    const float t7 = z2 + 7.0f * p2 - 4.0f;
    const float azmp = fabsf(zmp);
    const bool nz = azmp > 0.0f;
    const float r = nz ? rcpf_(azmp) : 0.0f;
    const float m3 = nz ? ((zmp_d < 0.0) ? -3.0f : 3.0f) : 0.0f;
    const float b = zpp * zpp;
    float le, ed, kc2, al, be, pp, a2, b2, scale;
    if (inside) {
        le = p2;
        ed = eta2;
        const float oma = f2 * f3;
        const float rso = rsqf_(oma);
        kc2 = d1 * f4 * rso * rso;
        al = 1.0f - 5.0f * z2 + p2 + q * q;
        be = oma * t7;
        pp = nz ? zpp * r : 1.0f;
        a2 = m3 * pp;
        b2 = m3;
        scale = (float)(2.0 / (9.0 * kPi)) * rso;
    } else {
        const float f14 = d1 * f4, f34 = f3 * f4;
        const float area4 = sqrtf_(f14 * (f2 * f3));
        const float kap0 = 2.0f * atan_pos_f((f14 > 0.0f) ? area4 * rcpf_(f14) : INFINITY);
        const float kap1 = 2.0f * atan_pos_f(area4 * rcpf_(f34));
        le = (p2 * kap0 + kap1 - 0.5f * area4) * (float)(1.0 / kPi);
        ed = (kap1 + 2.0f * eta2 * kap0 - 0.25f * (1.0f + 5.0f * p2 + z2) * area4) * (float)(1.0 / kTwoPi);
        const float fzp = 4.0f * zf * pf;
        const float rsz = rsqf_(fzp);
        kc2 = d1 * f4 * rsz * rsz;
        al = (1.0f - b) * (2.0f * b + a - 3.0f) - 3.0f * q * (b - 2.0f);
        be = fzp * t7;
        pp = nz ? r : 1.0f;
        b2 = m3 * zpp;
        a2 = b2 * r;
        scale = (float)(2.0 / (9.0 * kPi)) * rsz;
    }
    const float s = cel_pair_f(sqrtf_(kc2), al + be, fmaf(be, kc2, al), pp, a2, b2);
    const float ld = fmaf(scale, s, theta);
    const float deficit = (float)L.cle * le + (float)L.cld * ld + (float)L.ced * ed;
    return 1.0 - (double)deficit;
}

// ---------------------------------------------------------------------------------------
// Kepler's equation, full solve: bracket-safeguarded Halley on m = |M| in [0, pi].
// Returns (sin E, cos E).  The iteration stops one step early: a Halley step below 1e-6
// leaves an error ~ step^3, and (sin, cos) follow by a third-order update.
__device__ __forceinline__ void kepler_full(double M, double e, double& sE, double& cE)
{
    M = reduce_2pi(M);
    const double sgn = (M < 0.0) ? -1.0 : 1.0;
    const double m = fabs(M);
    double lo = m, hi = fmin(m + e, kPi);
    double s, c;
    sincos_red(m, s, c);
    double x = m;
    bool done = (e == 0.0);
    {
        const double x0 = m + e * s * __builtin_amdgcn_rsq(fma(e, e - 2.0 * c, 1.0));
        if (!done) x = (x0 >= lo && x0 <= hi) ? x0 : 0.5 * (lo + hi);
    }
#pragma unroll 1
    for (int it = 0; it < 30; ++it) {
        if (__all(done)) break;
        double sx, cx;
        sincos_red(x, sx, cx);
        const double f = x - e * sx - m;
        if (f > 0.0) hi = x; else lo = x;
        const double fp = 1.0 - e * cx;
        const double rfp = rcp_nr1(fp);
        double dx = -f * rcp_nr1(fma(-0.5 * f * rfp, e * sx, fp));
        double xn = x + dx;
        const bool safe = (xn >= lo && xn <= hi);
        if (!safe) { xn = 0.5 * (lo + hi); dx = xn - x; }
        if (!done) {
            x = xn;
            // (sin, cos) at x + dx from those at x
            const double h = dx * dx;
            s = fma(dx, fma(-h, 1.0 / 6.0, 1.0) * cx, fma(-0.5 * h, sx, sx));
            c = fma(-dx, fma(-h, 1.0 / 6.0, 1.0) * sx, fma(-0.5 * h, cx, cx));
        }
        done = done || (safe && fabs(dx) < 1e-6) || !(dx == dx);
    }
    sE = sgn * s;
    cE = c;
}

// Advance (sinE, cosE) of E - e sinE = M to M + dM for small dM: Newton on
// g(d) = d - e (sinE (cos d - 1) + cosE sin d) - dM with Taylor kernels and a running
// reciprocal of g' (no division).  Returns false when |d| is too large for the series.
// FREEZE: a lane that has settled keeps its numbers while the others go on, so that what a lane returns does
// not depend on which other lanes share its wave (the bounded evaluation changes that from run to run, and
// its results must not); without it every lane takes the steps the slowest one needs (2 % faster).
template <bool FREEZE = false>
__device__ __forceinline__ bool kepler_step(double dM, double e, double& sE, double& cE)
{
    double rho = rcp_nr1(fma(-e, cE, 1.0));
    double d = dM * rho;
    if (!(fabs(d) < 0.08)) return false;
    // second-order start: one Newton step then suffices (a third-order start, tried in round 3, changes
    // nothing: the lanes of a trip nearly always settle in one step already)
    d = fma(-0.5 * e * sE * rho * d, d, d);
    // (a lane that has settled keeps its numbers while the others go on: what a lane returns must not
    // depend on which other lanes share its wave -- the bounded evaluation changes that from run to run)
    double ds = 0.0, dc = 0.0, step = 0.0;
    bool settled = false;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        if (!FREEZE || !settled) {                       // (an exec-masked block: settled lanes keep their numbers)
            const double d2 = d * d;
            const double sd = d * fma(d2, fma_k(d2, fma_k(d2, fma_k(d2, 1.0 / 362880.0, -1.0 / 5040.0),
                                                        1.0 / 120.0), -1.0 / 6.0), 1.0);
            const double c1 = d2 * fma(d2, fma_k(d2, fma_k(d2, fma_k(d2, -1.0 / 3628800.0, 1.0 / 40320.0),
                                                         -1.0 / 720.0), 1.0 / 24.0), -0.5);
            ds = fma(sE, c1, cE * sd);   // sin(E+d) - sin E
            dc = fma(cE, c1, -sE * sd);  // cos(E+d) - cos E
            const double g = fma(-e, ds, d) - dM;
            const double gp = fma(-e, cE + dc, 1.0);
            rho = rho * fma(-gp, rho, 2.0);              // Newton on 1/gp, twice: what the last step (< 1e-9) leaves behind
            rho = rho * fma(-gp, rho, 2.0);              // is step x (rho's relative error), see kepler_step_wide
            step = g * rho;
            d -= step;
            // a step below 1e-9 leaves an error ~ step^2: done after applying it
            settled = fabs(step) < 1e-9;
        }
        if (__all(settled)) break;
    }
    // (ds, dc) were evaluated one step back: first-order correction
    const double s1 = sE + ds, c1 = cE + dc;
    sE = fma(-step, c1, s1);
    cE = fma(step, s1, c1);
    return true;
}

// The same step for |d| up to 0.3 rad (two more Taylor terms each): from the row's solution at
// inferior conjunction to an exposure centre inside the transit window -- a tenth of an orbit at most
// for the reference's periods -- instead of a full solve per cell (plan_cell), and to the points of the
// secondary-eclipse scan (rowc_kernel).
template <bool FREEZE = false>
__device__ __forceinline__ bool kepler_step_wide(double dM, double e, double& sE, double& cE)
{
    double rho = rcp_nr1(fma(-e, cE, 1.0));
    const double x = dM * rho;
    if (!(fabs(x) < 0.3)) return false;
#ifdef TRX_KEPLER_NEWTON
    // (rounds 1-5, A/B builds: second-order start, Newton steps, settled below 1e-9 -- two to three trips per plan)
    double d = fma(-0.5 * e * sE * rho * x, x, x);
#else
    // Third-order start: with sin d = d - d^3/6, cos d - 1 = -d^2/2 the equation reads d + a2 d^2 + a3 d^3 = x,
    // a2 = e sinE rho / 2, a3 = e cosE rho / 6, whose inverse series is d = x - a2 x^2 + (2 a2^2 - a3) x^3 + O(x^4).
    const double a2 = 0.5 * e * sE * rho, a3 = (1.0 / 6.0) * e * cE * rho;
    double d = x * fma(x, fma(x, fma(2.0 * a2, a2, -a3), -a2), 1.0);
#endif
    double ds = 0.0, dc = 0.0, step = 0.0;
    bool settled = false;
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
        if (!FREEZE || !settled) {
            const double d2 = d * d;
            const double sd = d * fma(d2, fma_k(d2, fma_k(d2, fma_k(d2, fma_k(d2, fma_k(d2, 1.0 / 6227020800.0,
                                  -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
            const double c1 = d2 * fma(d2, fma_k(d2, fma_k(d2, fma_k(d2, fma_k(d2, fma_k(d2, -1.0 / 87178291200.0,
                                  1.0 / 479001600.0), -1.0 / 3628800.0), 1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5);
            ds = fma(sE, c1, cE * sd);
            dc = fma(cE, c1, -sE * sd);
            const double g = fma(-e, ds, d) - dM;
            const double gp = fma(-e, cE + dc, 1.0);
            // Newton on 1 / g', twice: over a step of up to 0.3 rad g' moves by up to e x 0.3, one refinement leaves
            // rho that far off squared, and what the last step leaves behind is step x (rho's relative error) -- with
            // one refinement up to 1e-14 in E, which a wide orbit (a / R = 30) turns into several 1e-13 of flux where
            // the Mandel-Agol expressions are badly conditioned (z near k; found by profiles/fuzz_kernels.py,
            // profiles/r03/fuzz.txt).  The second refinement squares it away.
#ifdef TRX_KEPLER_NEWTON
            rho = rho * fma(-gp, rho, 2.0);
            rho = rho * fma(-gp, rho, 2.0);
            step = g * rho;
            d -= step;
            settled = fabs(step) < 1e-9;
#else
            // (1 / g' afresh, to 2^-48: what a step leaves behind includes step x (the reciprocal's relative error), and
            // with steps of up to 1e-6 counting as the last one, two refinements of the previous reciprocal -- 1e-4
            // relative after a first step across e rho d ~ 0.3 -- are no longer enough: 3e-10 in flux, found by the fuzz)
            rho = rcp_nr1(gp);
            // Halley's step (round 6): t / (1 - t h) = t + t^2 h + O(t^3 h^2) with t = g / g' and h = g'' / (2 g') =
            // e sin(E + d) / (2 g') -- sin(E + d) is at hand -- converges cubically: a step below 1e-6 leaves
            // (h^2 + e rho / 6) x 1e-18 behind even at e rho = 9, where a Newton step had to be below 1e-9 for the same.
            // With the third-order start a plan settles in ONE trip up to x ~ 0.05 (a transit of P / 60) and in two up
            // to the 0.3 this function takes; rounds 1-5 took two and three (profiles/r06/ab_kepler.txt).
            const double t = g * rho;
            const double h = 0.5 * e * (sE + ds) * rho;
            step = fma(t * t, h, t);
            d -= step;
            settled = fabs(step) < 1e-6;
#endif
        }
        if (__all(settled)) break;
    }
#ifdef TRX_KEPLER_NEWTON
    if (!(fabs(step) < 1e-7)) return false;       // (a lane that did not settle: full solve)
    const double s1 = sE + ds, c1 = cE + dc;
    sE = fma(-step, c1, s1);
    cE = fma(step, s1, c1);
#else
    if (!settled) return false;                   // (a lane that did not settle: full solve)
    // (ds, dc) were evaluated one step back; the step is up to 1e-6: second-order correction (the third order is 2e-19)
    const double s1 = sE + ds, c1 = cE + dc;
    const double hs = 0.5 * step * step;
    sE = fma(-hs, s1, fma(-step, c1, s1));
    cE = fma(-hs, c1, fma(step, s1, c1));
#endif
    return true;
}

// ---------------------------------------------------------------------------------------
// Per-row constants staged in LDS (one row = one Monte-Carlo draw).
struct RowC {
    double k, t0, nmot, e, Mtr;
    double ax, ay, bx, by, cosi;
    double wlo, whi;          // mean-anomaly window (relative to conjunction), margins included
    double cle, cld, ced;     // limb-darkening weights
    double rdil;              // dilution: 1 - m_out = (1 - m) * rdil, rdil = 1 / ((1 + xeb)(1 + fdil)) for the reference's
                              // m = (m + xeb)/(1 + xeb) [EB], m = (m + fdil)/(1 + fdil); NaN when a flux ratio is not finite
    double sEt, cEt;          // sin, cos of the eccentric anomaly at inferior conjunction (where M = Mtr)
    double excl;              // 1.0 when the EB secondary rule excludes the draw (+inf), else 0
};
constexpr int kRowDoubles = sizeof(RowC) / sizeof(double);

// Orbit constants from pytransit-shaped (k, t0, p, a, i, e, w).  Runs once per row.
template <bool WINDOW = true>
__device__ __forceinline__ void orbit_init(RowC& c, double k, double t0, double per, double a,
                                           double inc, double e, double w, double exptime)
{
    const double ftr = kHalfPi - w;
    const double rt = sqrt(1.0 - e * e);
    double sf, cf;
    sincos_red(ftr, sf, cf);
    const double Etr = atan2(rt * sf, e + cf);
    double sEt, cEt;
    sincos_red(Etr, sEt, cEt);
    double sw, cw;
    sincos_red(w, sw, cw);
    c.k = k;
    c.t0 = t0;
    c.nmot = kTwoPi / per;
    c.e = e;
    c.Mtr = Etr - e * sEt;
    c.sEt = sEt;
    c.cEt = cEt;
    c.ax = a * cw;
    c.bx = -a * rt * sw;
    c.ay = a * sw;
    c.by = a * rt * cw;
    double si;
    sincos_red(inc, si, c.cosi);
    if (!WINDOW) {
        // no window (the 25-point secondary-eclipse scan sits inside its eclipse window anyway):
        // every cell is evaluated
        c.wlo = -INFINITY;
        c.whi = INFINITY;
        return;
    }
    // Window: X(E) = ax (cosE - e) + bx sinE = A cos(E - phi) - ax e; transit needs |X| < 1+k.
    const double R = (1.0 + k) * (1.0 + 1e-9) + 1e-12;
    const double A = sqrt(c.ax * c.ax + c.bx * c.bx);
    const double phi = atan2(c.bx, c.ax);
    const double clo = (c.ax * e - R) / A, chi = (c.ax * e + R) / A;
    const double psi = remainder(Etr - phi, kTwoPi);
    double plo, phi2;  // arc [plo, phi2] of psi = E - phi containing psi
    // the arc of the strip |X| < r that holds conjunction; returns true in the case with a mirror arc (both ends closed)
    double xlo = 0.0, xhi = 0.0;
    auto strip_arc = [&](double clo_, double chi_, double& lo_, double& hi_) -> bool {
        const bool open_hi = !(chi_ < 1.0), open_lo = !(clo_ > -1.0);
        xlo = acos(fmin(fmax(clo_, -1.0), 1.0));
        xhi = acos(fmin(fmax(chi_, -1.0), 1.0));
        if (open_hi && open_lo) {
            lo_ = psi - kTwoPi;
            hi_ = psi + kTwoPi;
        } else if (open_hi) {
            lo_ = -xlo;
            hi_ = xlo;
        } else if (open_lo) {
            if (psi >= 0.0) { lo_ = xhi; hi_ = kTwoPi - xhi; } else { lo_ = xhi - kTwoPi; hi_ = -xhi; }
        } else {
            if (psi >= 0.0) { lo_ = xhi; hi_ = xlo; } else { lo_ = -xlo; hi_ = -xhi; }
            return true;
        }
        return false;
    };
    bool widened = false;
    const double Ay = sqrt(c.ay * c.ay + c.by * c.by);          // Y(E) = Ay cos(E - phy) - ay e
    const double phy = atan2(c.by, c.ay);
    if (strip_arc(clo, chi, plo, phi2)) {
        // The strip |X| < 1+k is met on a second arc, the mirror image about psi = 0.  It normally
        // lies on the far side of the orbit (Y < 0: no transit), but on a very eccentric orbit seen
        // nearly along its major axis its end next to psi = 0 can still be on the near side: a
        // second, grazing passage a few hours from conjunction (e = 0.9: flux deficit 6e-5,
        // found by tests/test_gpu_kernels.py on irregular time stamps).  Y(E) = Ay cos(E - phy) -
        // ay e is a sinusoid too: if its maximum over the mirror arc is not negative the window
        // becomes the hull of the two arcs (the cells in between are evaluated and come out as 1).
        const double o1 = (psi >= 0.0) ? -xlo : xhi, o2 = (psi >= 0.0) ? -xhi : xlo;
        const double d1 = remainder(Etr + (o1 - psi) - phy, kTwoPi), span = o2 - o1;
        const bool peak = (d1 <= 0.0 && d1 + span >= 0.0) || (d1 + span >= kTwoPi);
        double s1, c1, s2, c2;
        sincos_red(d1, s1, c1);
        sincos_red(d1 + span, s2, c2);
        const double cmax = peak ? 1.0 : fmax(c1, c2);
        if (!(Ay * cmax - c.ay * e < -1e-9 * (Ay + fabs(c.ay * e)))) {
            plo = fmin(plo, o1);
            phi2 = fmax(phi2, o2);
            widened = true;
        }
    }
    double Elo = Etr + (plo - psi), Ehi = Etr + (phi2 - psi);
    double sl, cl, sh, ch;
    sincos_red(Elo, sl, cl);
    sincos_red(Ehi, sh, ch);
#ifndef TRX_STRIP_WINDOW
    // From the strip to the DISC (round 6).  The strip |X| < 1 + k ignores the impact parameter: a transit's chord is
    // shorter than the star's diameter by sqrt(1 - b^2 / (1 + k)^2), pi/4 of it on average, and every cell of the
    // strip outside the disc was filed, planned -- a Kepler step and the node criteria, as dear as an evaluation -- and
    // found to need nothing (a third of the planned cells of BASELINE config 1: profiles/r06/window_census.txt).
    // On the arc, z^2 = X^2 + (Y cos i)^2 >= X^2 + (min |Y| cos i)^2, so the disc lies inside the NARROWER strip
    // |X| < sqrt(R^2 - (min |Y| cos i)^2): a superset of the cells that see the planet, by construction.  Y is a sinusoid
    // of E: over an arc shorter than pi its smallest value sits at an end, or at its trough when the arc holds it.  One
    // such step from the strip's ends gets within ~(R / a)^2 of the true contacts; it is not taken when the arc is the
    // hull of two passages, when Y is not positive over the whole arc, or when anything is NaN (every comparison false).
    if (!widened && (phi2 - plo) < kPi) {
        const double Yl = fma(c.ay, cl - e, c.by * sl), Yh = fma(c.ay, ch - e, c.by * sh);
        const double d1 = remainder(Elo - phy - kPi, kTwoPi), span = Ehi - Elo;
        const bool trough = (d1 <= 0.0 && d1 + span >= 0.0) || (d1 + span >= kTwoPi);
        const double Ymin = trough ? fmin(fmin(Yl, Yh), -Ay - c.ay * e) : fmin(Yl, Yh);
        if (Ymin > 0.0) {
            const double yc = Ymin * c.cosi * (1.0 - 1e-9);
            const double R2 = fma(-yc, yc, R * R);
            if (R2 > 0.0) {
                const double Rt = sqrt(R2);
                double lo2, hi2;
                strip_arc((c.ax * e - Rt) / A, (c.ax * e + Rt) / A, lo2, hi2);
                // (inside the arc it came from: the narrower strip's arc around conjunction cannot reach beyond it)
                lo2 = fmax(lo2, plo);
                hi2 = fmin(hi2, phi2);
                if (lo2 <= psi && psi <= hi2) {
                    plo = lo2;
                    phi2 = hi2;
                    Elo = Etr + (plo - psi);
                    Ehi = Etr + (phi2 - psi);
                    sincos_red(Elo, sl, cl);
                    sincos_red(Ehi, sh, ch);
                }
            } else if (R2 <= 0.0) {
                // the body passes the star by: no point of the arc is on the disc (an empty window: wlo > whi)
                c.wlo = 1e300;
                c.whi = -1e300;
                return;
            }
        }
    }
#endif
    const double mg = 0.5 * fabs(c.nmot * exptime) * (1.0 + 1e-9) + 1e-11;
    c.wlo = (Elo - Etr) - e * (sl - sEt) - mg;
    c.whi = (Ehi - Etr) - e * (sh - sEt) + mg;
    // NaN anywhere (invalid draw) => the comparisons in in_window are false => the cell is
    // evaluated and the NaN propagates to the result exactly as in the plain algorithm.
}

__device__ __forceinline__ bool in_window(double wlo, double whi, double dMc)
{
    // dMc in [-pi, pi]: mean anomaly of the exposure centre relative to conjunction
    const bool out = ((dMc < wlo) && !(dMc + kTwoPi <= whi)) ||
                     ((dMc > whi) && !(dMc - kTwoPi >= wlo));
    return !out;
}

// Reduced node sets for the exposure average (filled on the host for the launch's S).
// The reference averages the model over S equally spaced sub-exposures.  Where the model is an
// analytic function of time over the exposure, that average is reproduced to ~1e-14 by the
// n-point Gauss rule of that very measure (S points of weight 1/S), n << S: a fixed weighted sum
// of n model values, exact for polynomials of degree 2n-1.  The model is analytic away from the
// limb contacts z = 1 + k and z = |1 - k|; the rule's error falls geometrically with the
// distance to the nearest (real or complex) contact time in units of the half exposure
// (profiles/r01/q_tier_error.txt), so each tier carries the zero-free radius it needs.
constexpr int kTiers = 7, kTierMaxNodes = 10;
struct TierTable {
    int n[kTiers];                         // nodes per tier, ascending; 0 = tier unused
    double radius[kTiers];                 // required contact-free radius / half exposure
    double x[kTiers * kTierMaxNodes];      // node offsets / exptime, ascending in time
    double w[kTiers * kTierMaxNodes];      // weights of the flux deficits 1 - f
};
// What of the table rides in a kernel's argument block: the node counts and radii (read with constant indices only,
// so a kernel that patches a local copy of its arguments keeps them in scalar registers).  The nodes and weights --
// indexed by a lane -- live in device memory, [kTiers * kTierMaxNodes] (offset, weight) pairs, staged in LDS by
// cells_kernel.
struct TierHead {
    int n[kTiers];
    double radius[kTiers];
};

// What one (row, time) cell has to evaluate: n nodes of tier `tier` (-1 = all S sub-exposures),
// or nothing (n = 0: the exposure is unocculted and its mean flux is exactly 1).
struct CellPlan {
    int n = 0, tier = -1;
    double sE = 0.0, cE = 1.0, Mprev = 0.0;     // eccentric-anomaly state carried along the nodes
    bool anchored = false;                       // state holds a solution (at the exposure centre)
    bool st_ok = false;                          // no limb contact within the stencil radius (cells_kernel)
    bool lazy = false;                           // the tier has not been looked for (plan_cell<..., LAZY>): see plan_tiers
};

// th_lds: the table's radii [kTiers] and node counts [kTiers] (as doubles) in LDS, or null.  cells_kernel stages them
// there: as kernel arguments they occupy 21 scalar registers for the whole kernel, which is already past the 102 it has
// (the compiler then parks scalars in VGPR lanes -- v_writelane / v_readlane, VALU issue both -- inside the chunk loop);
// an LDS read at a wave-uniform or lane-chosen address costs an LDS slot, no VALU issue.
constexpr int kTierHeadDoubles = 2 * kTiers;
// LAZY (the stencil instantiation of cells_kernel, st_radius > 0): a cell that passes the test at the stencil's radius
// will most likely take its exposure average from its neighbours' centre values and never need a tier; when every
// active lane of the wave passes it (or is off the disc altogether) the search for the tier -- three more tests and
// their selects, a fifth of this function -- is left out and the plan says so (CellPlan::lazy); plan_tiers() supplies
// it for the few cells that turn out to need their own nodes after all (a chunk's first and last cells).  Passing at
// the stencil's radius (>= 2.2 half exposures) implies passing at the smallest tier's (1.8): such a cell is never a
// contact cell.
template <bool CHECK_WINDOW = true, bool FREEZE = false, bool LDS_HEAD = false, bool LAZY = false>
__device__ __forceinline__ CellPlan plan_cell(const RowC& c, double t, double exptime, int S,
                                              const TierHead& tt, bool use_tiers, double st_radius = 0.0,
                                              const double* th_lds = nullptr)
{
    CellPlan p;
    const double phase = c.nmot * (t - c.t0);
    if (CHECK_WINDOW) {       // cells_kernel has filed the in-window cells already
        const double dMc = reduce_2pi(phase);
        // the reduction is good to ~1e-16 |phase|: widen the window by that much
        const double slack = 1e-15 * fabs(phase);
        if (!in_window(c.wlo - slack, c.whi + slack, dMc)) return p;
    }
    p.n = S;
    if (!use_tiers) return p;
    // orbit at the exposure centre: position, velocity and the quadratic model of z^2(t)
    const double opp = 1.0 + c.k, opp2 = opp * opp, omk = 1.0 - c.k;
    p.Mprev = phase + c.Mtr;
#ifndef TRX_PLAN_FULL_SOLVE
    // the in-window cells sit within a fraction of a radian of inferior conjunction: Newton steps from
    // the row's solution there (RowC::sEt, cEt) instead of a full solve; wave-uniform fallback
    p.sE = c.sEt;
    p.cE = c.cEt;
    const bool stepped = kepler_step_wide<FREEZE>(reduce_2pi(phase), c.e, p.sE, p.cE);
    if (!__all(stepped)) {
        double sF, cF;
        TRX_CENSUS_ADD(kCenKeplerFullPlan, 1);
        kepler_full(p.Mprev, c.e, sF, cF);
        if (!stepped) { p.sE = sF; p.cE = cF; }
    }
#else
    kepler_full(p.Mprev, c.e, p.sE, p.cE);
#endif
    p.anchored = true;
    const double rho = rcp_fast(fma(-c.e, p.cE, 1.0));
    const double ce = p.cE - c.e;
    const double X = fma(c.ax, ce, c.bx * p.sE), Y = fma(c.ay, ce, c.by * p.sE);
    const double yc = Y * c.cosi;
    const double nr = c.nmot * rho;
    const double Xp = fma(c.bx, p.cE, -c.ax * p.sE) * nr, Yp = fma(c.by, p.cE, -c.ay * p.sE) * nr;
    const double ycp = Yp * c.cosi;
    const double z2 = fma(X, X, yc * yc);
    const double g1 = fabs(2.0 * fma(X, Xp, yc * ycp));
    const double g2 = fabs(fma(Xp, Xp, ycp * ycp) - (nr * nr) * rho * z2);
    const double G = fmin(fabs(z2 - opp2), fabs(z2 - omk * omk));   // distance of z^2 to the contacts
    // the rule's error is relative to the eclipse depth (~k^2 up to total eclipses): deep
    // eclipses get up to 1.5x the radius so that the ABSOLUTE flux error stays below ~1e-13
    const double kd = fmin(c.k, 1.0);
    const double hx = 0.5 * fabs(exptime) * fma(0.5 * kd, kd, 1.0);
    const double om = fabs(nr) * rho;                                // bound on the angular rate
    // the smallest admissible node count wins.  The test is monotone in the radius and the radii fall with the
    // tier index, so with all seven tiers usable (S >= 20) three tests find it by bisection instead of seven
    auto admissible = [&](double radius) -> bool {
        const double tau = radius * hx;
        return G >= 1.25 * fma(g2 * tau, tau, g1 * tau) && om * tau <= 0.15 && Y > fabs(Yp) * tau;
    };
    static_assert(kTiers == 7, "bisection over seven tiers");
    if (LAZY && st_radius > 0.0) {
        p.st_ok = admissible(st_radius);
        const double tau1z = 1.5 * hx;
        const bool off = z2 > opp2 && G >= 1.25 * fma(g2 * tau1z, tau1z, g1 * tau1z) && om * tau1z <= 0.15;
        if (!__any(!p.st_ok && !off)) {
            p.lazy = true;
            if (off) p.n = 0;
            return p;
        }
    }
    // (the table's entries as VALUES, read with literal indices before any choice is made: `c ? tt.radius[1] : tt.radius[5]`
    // is a choice between two ADDRESSES to the compiler, and a loop over q an indexed one -- either keeps a kernel's
    // patched local copy of its argument block (star_args, trx_cells.hpp) from being split into registers, and the
    // whole block then lives in scratch memory)
    if (LDS_HEAD) {
        // (the table in LDS: the bisection reads three radii and one count at addresses it computes)
        if (th_lds[kTiers + kTiers - 1] > 0.0) {
            const bool ok3 = admissible(th_lds[3]);
            const bool ok1 = admissible(th_lds[ok3 ? 1 : 5]);
            const bool ok2 = admissible(th_lds[ok3 ? (ok1 ? 0 : 2) : (ok1 ? 4 : 6)]);
            const int q = (ok3 ? (ok1 ? 0 : 2) : (ok1 ? 4 : 6)) + (ok2 ? 0 : 1);      // first admissible tier, 7 = none
            if (q < kTiers) {
                p.tier = q;
                p.n = (int)th_lds[kTiers + q];
            }
        } else {
            for (int q = kTiers - 1; q >= 0; --q) {
                const bool ok = th_lds[kTiers + q] > 0.0 && admissible(th_lds[q]);
                if (ok) { p.tier = q; p.n = (int)th_lds[kTiers + q]; }
            }
        }
        if (st_radius > 0.0) p.st_ok = admissible(st_radius);
        const double tau1l = 1.5 * hx;
        if (z2 > opp2 && G >= 1.25 * fma(g2 * tau1l, tau1l, g1 * tau1l) && om * tau1l <= 0.15) p.n = 0;
        return p;
    }
    const int n0 = tt.n[0], n1 = tt.n[1], n2 = tt.n[2], n3 = tt.n[3], n4 = tt.n[4], n5 = tt.n[5], n6 = tt.n[6];
    const double r0 = tt.radius[0], r1 = tt.radius[1], r2 = tt.radius[2], r3 = tt.radius[3], r4 = tt.radius[4],
                 r5 = tt.radius[5], r6 = tt.radius[6];
    if (n6 > 0) {
        const bool ok3 = admissible(r3);
        const bool ok1 = admissible(ok3 ? r1 : r5);
        const bool ok2 = admissible(ok3 ? (ok1 ? r0 : r2) : (ok1 ? r4 : r6));
        const int q = (ok3 ? (ok1 ? 0 : 2) : (ok1 ? 4 : 6)) + (ok2 ? 0 : 1);      // first admissible tier, 7 = none
        if (q < kTiers) {
            p.tier = q;
            p.n = (q < 4) ? ((q < 2) ? (q == 0 ? n0 : n1) : (q == 2 ? n2 : n3))
                          : ((q < 6) ? (q == 4 ? n4 : n5) : n6);
        }
    } else {
        // scanned from the largest node count down
#define TRX_TIER_TRY(q, nq, rq) { const bool ok = nq > 0 && admissible(rq); if (ok) { p.tier = q; p.n = nq; } }
        TRX_TIER_TRY(6, n6, r6) TRX_TIER_TRY(5, n5, r5) TRX_TIER_TRY(4, n4, r4) TRX_TIER_TRY(3, n3, r3)
        TRX_TIER_TRY(2, n2, r2) TRX_TIER_TRY(1, n1, r1) TRX_TIER_TRY(0, n0, r0)
#undef TRX_TIER_TRY
    }
    // the same test at the radius the centre-value stencil of a dense uniform time grid needs
    if (st_radius > 0.0) p.st_ok = admissible(st_radius);
    // whole exposure off the disc: every sub-exposure is exactly 1
    const double tau1 = 1.5 * hx;
    if (z2 > opp2 && G >= 1.25 * fma(g2 * tau1, tau1, g1 * tau1) && om * tau1 <= 0.15) p.n = 0;
    return p;
}

// The tier of a cell planned with LAZY whose exposure-centre solution (p.sE, p.cE) is at hand: the same quantities and
// the same search as plan_cell, so the same tier.  Only for cells that passed the test at the stencil's radius (a tier
// exists).
__device__ __forceinline__ void plan_tiers(const RowC& c, double exptime, CellPlan& p, const double* th_lds)
{
    const double opp = 1.0 + c.k, opp2 = opp * opp, omk = 1.0 - c.k;
    const double rho = rcp_fast(fma(-c.e, p.cE, 1.0));
    const double ce = p.cE - c.e;
    const double X = fma(c.ax, ce, c.bx * p.sE), Y = fma(c.ay, ce, c.by * p.sE);
    const double yc = Y * c.cosi;
    const double nr = c.nmot * rho;
    const double Xp = fma(c.bx, p.cE, -c.ax * p.sE) * nr, Yp = fma(c.by, p.cE, -c.ay * p.sE) * nr;
    const double ycp = Yp * c.cosi;
    const double z2 = fma(X, X, yc * yc);
    const double g1 = fabs(2.0 * fma(X, Xp, yc * ycp));
    const double g2 = fabs(fma(Xp, Xp, ycp * ycp) - (nr * nr) * rho * z2);
    const double G = fmin(fabs(z2 - opp2), fabs(z2 - omk * omk));
    const double kd = fmin(c.k, 1.0);
    const double hx = 0.5 * fabs(exptime) * fma(0.5 * kd, kd, 1.0);
    const double om = fabs(nr) * rho;
    auto admissible = [&](double radius) -> bool {
        const double tau = radius * hx;
        return G >= 1.25 * fma(g2 * tau, tau, g1 * tau) && om * tau <= 0.15 && Y > fabs(Yp) * tau;
    };
    if (th_lds[kTiers + kTiers - 1] > 0.0) {
        const bool ok3 = admissible(th_lds[3]);
        const bool ok1 = admissible(th_lds[ok3 ? 1 : 5]);
        const bool ok2 = admissible(th_lds[ok3 ? (ok1 ? 0 : 2) : (ok1 ? 4 : 6)]);
        const int q = (ok3 ? (ok1 ? 0 : 2) : (ok1 ? 4 : 6)) + (ok2 ? 0 : 1);
        if (q < kTiers) {
            p.tier = q;
            p.n = (int)th_lds[kTiers + q];
        }
    } else {
        for (int q = kTiers - 1; q >= 0; --q) {
            const bool ok = th_lds[kTiers + q] > 0.0 && admissible(th_lds[q]);
            if (ok) { p.tier = q; p.n = (int)th_lds[kTiers + q]; }
        }
    }
    p.lazy = false;
}

// A cell already known to need all S sub-exposures (the contact cells' second sweep): only the exposure
// centre's solution, none of the criteria.
template <bool FREEZE = false>
__device__ __forceinline__ CellPlan plan_all_subexposures(const RowC& c, double t, int S)
{
    CellPlan p;
    p.n = S;
    const double phase = c.nmot * (t - c.t0);
    p.Mprev = phase + c.Mtr;
    p.sE = c.sEt;
    p.cE = c.cEt;
    const bool stepped = kepler_step_wide<FREEZE>(reduce_2pi(phase), c.e, p.sE, p.cE);
    if (!__all(stepped)) {
        double sF, cF;
        kepler_full(p.Mprev, c.e, sF, cF);
        if (!stepped) { p.sE = sF; p.cE = cF; }
    }
    p.anchored = true;
    return p;
}

// Node s (1-based) of the plan: advances the orbit; returns z^2 (NaN propagates) and Y (< 0 on
// the far side of the orbit).  frac = node offset / exptime.
__device__ __forceinline__ double node_z2(const RowC& c, CellPlan& p, double t, double exptime,
                                          double frac, bool stepping, double& Y)
{
    const double M = c.nmot * ((t + exptime * frac) - c.t0) + c.Mtr;
    bool have = false;
    if (stepping && p.anchored) {
        double sT = p.sE, cT = p.cE;
        have = kepler_step_wide(reduce_2pi(M - p.Mprev), c.e, sT, cT);
        if (have) { p.sE = sT; p.cE = cT; }
    }
    if (!have) kepler_full(M, c.e, p.sE, p.cE);
    p.anchored = true;
    p.Mprev = M;
    const double ce = p.cE - c.e;
    const double X = fma(c.ax, ce, c.bx * p.sE);
    Y = fma(c.ay, ce, c.by * p.sE);
    const double yc = Y * c.cosi;
    return fma(X, X, yc * yc);
}

template <bool FP32, int TAB = 1>
__device__ __forceinline__ double disc_flux(double z, double k, const Limb& L, const double* atab)
{
    return FP32 ? ma_flux_f32(z, k, L) : ma_flux<TAB>(z, k, L, atab);
}

// Mean model flux of one exposure (centre t) evaluated by one lane: the body of pytransit's
// evaluate_pv for one (row, time) cell.  Used for the 25-point secondary-eclipse scan (S = 1);
// the time axis of the light curve goes through the packed path of rows_kernel.
__device__ __forceinline__ double exposure_flux(const RowC& c, const Limb& L, double t,
                                                double exptime, int S, double dS, double rS, bool stepping,
                                                const TierHead& tt)
{
    CellPlan p = plan_cell(c, t, exptime, S, tt, false);
    if (p.n == 0) return 1.0;
    const double opp = 1.0 + c.k, opp2 = opp * opp;
    double acc = 0.0;
#ifndef TRX_PLAN_FULL_SOLVE
    // start from the row's solution at conjunction (node_z2 steps from it when the point is near enough)
    p.sE = c.sEt; p.cE = c.cEt; p.Mprev = c.Mtr; p.anchored = true;
    stepping = true;
#endif
#pragma unroll 1
    for (int s = 1; s <= S; ++s) {
        double Y;
        // exptime*((s-0.5)/S - 0.5) up to an ulp of the offset (~1e-20 d)
        const double z2 = node_z2(c, p, t, exptime, fma((double)s - 0.5, rS, -0.5), stepping, Y);
        double f = 1.0;
        if (Y >= 0.0 && z2 < opp2) f = ma_flux(sqrt_fast(z2), c.k, L);
        else if (z2 != z2) f = z2;
        acc += f;
    }
    return acc / dS;
}

// ---------------------------------------------------------------------------------------
// log-mean-exp partial state: running max m (finite or -inf), s = sum exp(x - m), pinf = saw +inf.
// NaN and -inf carry zero weight (_numerics.py:48).
struct Lme {
    double m, s;
    int pinf;
};

__device__ __forceinline__ void lme_merge(Lme& a, const Lme& b)
{
    a.pinf |= b.pinf;
    if (b.m == -INFINITY) return;
    if (a.m == -INFINITY) { a.m = b.m; a.s = b.s; return; }
    if (b.m > a.m) { a.s = fma(a.s, exp(a.m - b.m), b.s); a.m = b.m; }
    else           { a.s = fma(b.s, exp(b.m - a.m), a.s); }
}

// blocks a vector of n log-weights is reduced by (the partition decides the bits of the sum)
__host__ __device__ inline int lme_blocks(long n)
{
    const long want = (n + 256L * 8 - 1) / (256L * 8);
    return (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

// torch.argmin's order: NaN before everything, then the smallest value; ties -> the lowest index
__device__ __forceinline__ bool argmin_before(double a, long ia, double b, long ib)
{
    const bool na = a != a, nb = b != b;
    if (na != nb) return na;
    if (!na && a != b) return a < b;
    return ia < ib;
}

// ---------------------------------------------------------------------------------------
// The end of one lnZ_* branch of calc_probs (trx_scenario_enqueue): run by ONE wave -- the first wave of the block of
// lme_partial_kernel<SCEN> that finishes last -- from the per-block partials of that launch:
//   the evidence (the fold of lme_final_kernel: lane l takes partials l, l + 64, ... in order, then a fixed butterfly),
//   the first minimum of chi^2 (NaN first, ties to the lowest index: numpy's / torch's argmin), and the record
//     res[0 .. ncol)  the best draw's columns (draw 0 when no draw passed the mask)
//     res[ncol]       lnZ          res[ncol + 1]  the masked count
//     res[16]         rows that hold the smallest chi^2 (> 1: the best draw is the first of an exact tie)
//     res[17]         status: 1 = a row of the branch was never written by any likelihood pass (lnZ is NaN then)
//     flag_out[0]     the limb-darkening flag of the draw kernel (branch 0 writes it)
// `res` may be pinned host memory (the record then needs no copy).  `state` is the persistent block at the head of
// the stream's scenario scratch: [0] the counter of finished blocks, [1] the draw kernel's flag -- both are left at
// zero for the next call (the last branch of a call clears the flag).
constexpr int kScenRecord = 18;       // TRX_SCENARIO_OUT
constexpr int kScenTies = 16, kScenStatus = 17;
struct ScenFinal {
    const int* idx;                   // the branch's list of masked draws
    const double* cols;               // [ncol][N]
    const double* cols0;              // dense: [ncol][cols0_stride] columns of draw 0 (the record when no draw passed the mask)
    int cols0_stride;                 // 0 or 1: [ncol]; K: the block also holds draws 1 .. K - 1 (a table's spare rows)
    int dense;                        // the masked draws' columns sit densely in `cols` (compact_fill_kernel): row r at
                                      // position r (branch 1: N - 1 - r); 0: at their draw index idx[r]
    long N, n_total;
    int ncol, branch, last_branch;
    double* res;                      // this branch's record
    double* flag_out;                 // where the flag goes (as a double), or null: branch 0 reads it and clears it
    unsigned* state;                  // persistent: [0] finished blocks of THIS branch, [1] the call's flag (branch 0's block)
};

__device__ __forceinline__ void scenario_final(const ScenFinal& f, const double* __restrict__ w,
                                               const double* __restrict__ pv, const long* __restrict__ pi,
                                               const long* __restrict__ pc, const long n, const int lane)
{
    const int nparts = lme_blocks(n);
    Lme t{-INFINITY, 0.0, 0};
    double bv = INFINITY;
    long bi = -1, bc = 0;
    if (n > 0) {
        for (int i = lane; i < nparts; i += 64) {
            Lme o{w[3 * i], w[3 * i + 1], (int)w[3 * i + 2]};
            lme_merge(t, o);
            const long oi = pi[i];
            if (oi >= 0) {
                if (bi < 0) { bv = pv[i]; bi = oi; bc = pc[i]; }
                else if (pv[i] == bv || (pv[i] != pv[i] && bv != bv)) { bc += pc[i]; bi = oi < bi ? oi : bi; }
                else if (argmin_before(pv[i], oi, bv, bi)) { bv = pv[i]; bi = oi; bc = pc[i]; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Lme other;
        other.m = __shfl_xor(t.m, o, 64);
        other.s = __shfl_xor(t.s, o, 64);
        other.pinf = __shfl_xor(t.pinf, o, 64);
        lme_merge(t, other);
        const double ov = __shfl_xor(bv, o, 64);
        const long oi = __shfl_xor(bi, o, 64);
        const long oc = __shfl_xor(bc, o, 64);
        if (oi >= 0) {
            if (bi < 0) { bv = ov; bi = oi; bc = oc; }
            else if (ov == bv || (ov != ov && bv != bv)) { bc += oc; bi = oi < bi ? oi : bi; }
            else if (argmin_before(ov, oi, bv, bi)) { bv = ov; bi = oi; bc = oc; }
        }
    }
    double lnz;
    const bool unwritten = (t.pinf & 2) != 0;                     // a row no likelihood pass wrote (rowc_kernel's mark)
    if (unwritten) lnz = NAN;
    else if (t.pinf & 1) lnz = INFINITY;                          // _numerics.py:46-47
    else if (t.m == -INFINITY) lnz = -INFINITY;                   // :49-50
    else lnz = log(t.s) + t.m - log((double)f.n_total);           // :51
    if (f.dense) {
        const long best = f.branch ? f.N - 1 - bi : bi;
        if (lane < f.ncol) f.res[lane] = (bi >= 0) ? f.cols[(long)lane * f.N + best] : f.cols0[(long)lane * (f.cols0_stride > 1 ? f.cols0_stride : 1)];
    } else {
        const long best = (bi >= 0) ? (long)f.idx[bi] : 0;
        if (lane < f.ncol) f.res[lane] = f.cols[(long)lane * f.N + best];
    }
    if (lane == f.ncol) f.res[f.ncol] = lnz;
    if (lane == f.ncol + 1) f.res[f.ncol + 1] = (double)n;
    if (lane == 62) {
        f.res[kScenTies] = (double)bc;
        f.res[kScenStatus] = unwritten ? 1.0 : 0.0;
    }
    if (lane == 63) {
        // (the flag belongs to the call: the branch that reports it -- branch 0 -- also clears it; with the branches of a
        // call reduced side by side in one launch the other branch must not touch it)
        if (f.flag_out) { f.flag_out[0] = (double)f.state[1]; f.state[1] = 0u; }
        f.state[0] = 0u;
    }
}

}  // namespace trx
