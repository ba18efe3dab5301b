// libtrx.so, part 2: the per-draw half of a scenario evidence.
//
// Every lnZ_* of the reference (marginal_likelihoods.py:39-2362) turns N uniform draws into
// per-draw stellar and orbital columns, a geometry mask and a companion prior through ~25 length-N
// numpy temporaries (and, in lnZ_STP / lnZ_SEB, a Python loop over N for the limb-darkening
// lookup): priors.py:16-383 (inverse-CDF samplers), funcs.py:54-140 (stellar and flux relations),
// priors.py:580-1005 (bound-companion and background priors), marginal_likelihoods.py:101-123
// (transit probability, collision and inclination masks).  draw_kernel does all of it for one
// draw per thread, for any of the ten scenarios, from staged random numbers (torch's generator on
// the device, or numpy's global stream copied over in the reference's draw order)
// or from its own Philox4x32-10 blocks, and writes the SoA parameter block trx_lnl_batch consumes plus the
// mask(s) and lnprior.  The host prepares nothing per draw: only the constants of the broken power laws, the
// spline coefficients and the small lookup tables (trx_draw_args, include/trx.h).
//
// One body, draw_one<PHASE>, three uses: trx_draw_scenario evaluates every draw in full (draw_kernel<0>);
// trx_scenario_enqueue first takes the geometry mask(s) of all draws (draw_kernel<1>, or <2> behind an fp32
// pre-test of the geometry that leaves the fp64 mask to the draws near it) and, once the survivors are listed,
// the columns and the prior of those alone (compact_fill_kernel) -- nine draws in ten fail the geometry.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/trx.h"
#include "trx_internal.hpp"

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kG = 6.6743e-08, kMsun = 1.988409870698051e+33, kRsun = 69570000000.0,
                 kRearth = 637810000.0, kAu = 14959787070000.0;

// ---- counter-based random numbers ------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror & Shaw 2011): ten rounds of two 32x32 -> 64 multiplies (one v_mad_u64_u32
// each) and three xors -- ~70 instructions a block.  (Threefry4x32-12 from the same paper -- adds, rotates and
// xors only -- was tried on the assumption that the integer multiplies run at a quarter rate here: ~90
// instructions a block, and the draw kernel got 8 % slower.)  Counter = (draw index lo, hi, slot, sub-draw),
// key = the call's seed.
struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        const U4 n{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// one block of 128 random bits for (seed, draw index, slot, sub-draw)
__device__ __forceinline__ U4 random_block(unsigned long long seed, long i, unsigned slot, unsigned sub)
{
    return philox4x32_10(U4{(unsigned)i, (unsigned)((unsigned long long)i >> 32), slot, sub},
                         (unsigned)seed, (unsigned)(seed >> 32));
}

// two uniforms in [0, 1) with 53 random bits each (numpy's construction: (a >> 5) 2^26 + (b >> 6))
__device__ __forceinline__ void philox_uniform2(unsigned long long seed, long i, unsigned slot, unsigned sub,
                                                double& u0, double& u1)
{
    const U4 r = random_block(seed, i, slot, sub);
    u0 = ((double)(r.x >> 5) * 67108864.0 + (double)(r.y >> 6)) * (1.0 / 9007199254740992.0);
    u1 = ((double)(r.z >> 5) * 67108864.0 + (double)(r.w >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double philox_uniform(unsigned long long seed, long i, unsigned slot)
{
    double u0, u1;
    philox_uniform2(seed, i, slot, 0u, u0, u1);
    return u0;
}

// x ** y for x >= 0 as exp(y ln x): ~|y ln x| ulp (a few 1e-15 relative here) instead of libm's
// correctly rounded pow, whose double-double arithmetic is a third of this kernel's instructions
__device__ __forceinline__ double pow_pos(double x, double y) { return exp(y * log(x)); }

// The planets' eccentricity, Beta(0.867, 3.030) (priors.py:146-148), as X / (X + Y) of two gamma variates by
// Marsaglia & Tsang (2000); Gamma(0.867) through Gamma(1.867) U^(1/0.867).  ONE counter block per attempt
// serves both variates: 32 bits for the radius and 32 for the angle of a Box-Muller pair -- its cosine branch
// proposes for X, its sine branch for Y -- and 24 bits for each acceptance test; an attempt is repeated (next
// block) for the variate(s) it did not settle.  The variate is a random DRAW, not a model quantity: its
// arithmetic runs in fp32 on the hardware's log / cos / sqrt / exp (relative error ~1e-7, a shift of the sampled
// distribution far below what 1e6 draws resolve; the fp64 libm versions were a third of a planet scenario's draw
// kernel).  The radius' argument keeps its fp64 exponent (>= 2^-33): the normals reach 6.7 sigma.
__device__ __forceinline__ double random_ecc_beta(unsigned long long seed, long i)
{
    const float dx = 1.867f - 1.0f / 3.0f, dy = 3.030f - 1.0f / 3.0f;
    const float cx = 1.0f / sqrtf(9.0f * dx), cy = 1.0f / sqrtf(9.0f * dy);
    float gx = dx, gy = dy;
    bool okx = false, oky = false;
    for (unsigned att = 0; att < 64u && !(okx && oky); ++att) {
        const U4 r = random_block(seed, i, 8u, att + 1u);
        const float rad = sqrtf(-2.0f * __logf((float)(((double)r.x + 0.5) * (1.0 / 4294967296.0))));
        const float ang = 6.2831853071795865f * ((float)r.y * (1.0f / 4294967296.0f));
        const float ua = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f);       // in (0, 1)
        const float ub = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
        if (!okx) {
            const float x = rad * __cosf(ang), t = 1.0f + cx * x;
            const float v = t * t * t;
            if (t > 0.0f && __logf(ua) < 0.5f * x * x + dx * (1.0f - v + __logf(v))) { gx = dx * v; okx = true; }
        }
        if (!oky) {
            const float y = rad * __sinf(ang), t = 1.0f + cy * y;
            const float v = t * t * t;
            if (t > 0.0f && __logf(ub) < 0.5f * y * y + dy * (1.0f - v + __logf(v))) { gy = dy * v; oky = true; }
        }
    }
    // sub-draw 0: the uniform that takes Gamma(1.867) to Gamma(0.867)
    const float boost = (float)(1.0 - philox_uniform(seed, i, 8u));
    gx *= __expf(__logf(boost) * (1.0f / 0.867f));
    return (double)(gx / (gx + gy));
}

// ---- tables staged in LDS ------------------------------------------------------------------
#ifndef TRX_DRAW_CHUNK
#define TRX_DRAW_CHUNK 1024
#endif
constexpr int kDrawChunk = TRX_DRAW_CHUNK;        // draws a workgroup pre-tests before it regroups the candidates

// 1 (round 6): the planets' eccentricity by inverse CDF from the spare half of the (5, 6) counter block; 0: the
// Marsaglia-Tsang sampler of rounds 3-5 (random_ecc_beta: two more blocks and six fp32 transcendentals per attempt --
// ~250 of the ~850 instructions a planet scenario's pre-tested draw costs)
#ifndef TRX_ECC_ICDF
#define TRX_ECC_ICDF 1
#endif
#include "trx_ecc_icdf.inc"

struct Tables {
    double spl[TRX_DRAW_N_SPLINES][TRX_DRAW_SPLINE_DOUBLES];
    double cc_sep[TRX_DRAW_MAX_CC], cc_con[TRX_DRAW_MAX_CC];
    double lut[2][TRX_DRAW_MAX_LUT];
    float ecc_lo[kEccIcdfN + 1], ecc_hi[kEccIcdfN + 1];      // inverse CDF of Beta(0.867, 3.030): see ecc_from_uniform
};

// The planets' eccentricity, Beta(0.867, 3.030) (priors.py:146-148), by inverse CDF: F^-1 tabulated in the variables in
// which it is smooth at its ends -- w = u^(1/a) below the median (e ~ w near 0), v = (1 - u)^(1/b) above it (1 - e ~ v
// near 1) -- 256 intervals each, linear interpolation, fp32 arithmetic: within 1e-6 of scipy's betaincinv
// (profiles/r06/make_ecc_icdf.py, which generates the table), a shift of the sampled distribution far below what 1e6
// draws resolve -- the same standard the fp32 gamma sampler before it was held to.  One uniform, ~20 instructions.
__device__ __forceinline__ double ecc_from_uniform(const Tables& T, double u)
{
    const bool lo = u < 0.5;
    const float base = lo ? (float)u : (float)(1.0 - u);
    const float x = __builtin_amdgcn_exp2f((lo ? (1.0f / 0.867f) : (1.0f / 3.030f)) * __builtin_amdgcn_logf(base));   // 0 -> 0
    const float f = x * (lo ? (float)kEccIcdfN / kEccIcdfWA : (float)kEccIcdfN / kEccIcdfVB);
    int j = (int)f;
    j = j < 0 ? 0 : (j > kEccIcdfN - 1 ? kEccIcdfN - 1 : j);
    const float fr = f - (float)j;
    const float* tab = lo ? T.ecc_lo : T.ecc_hi;
    const float e0 = tab[j], e1 = tab[j + 1];
    return (double)fmaf(fr, e1 - e0, e0);
}

// piecewise cubic: i = (number of knots <= v) - 1 clamped, ((c0 d + c1) d + c2) d + c3
__device__ __forceinline__ double spline_eval(const double* s, double v)
{
    const int m = (int)s[0];
    const double* x = s + 1;
    int i = 0;
    for (int k = 1; k < m; ++k) i += (x[k] <= v) ? 1 : 0;
    const double d = v - x[i];
    const double* c = x + TRX_DRAW_MAX_KNOTS;
    return ((c[i] * d + c[TRX_DRAW_MAX_KNOTS + i]) * d + c[2 * TRX_DRAW_MAX_KNOTS + i]) * d +
           c[3 * TRX_DRAW_MAX_KNOTS + i];
}

// funcs.stellar_relations (funcs.py:54-79)
__device__ __forceinline__ void stellar_relations(const Tables& T, double M, double maxR, double maxT,
                                                  double& R, double& Te)
{
    const bool hot = M > 0.63;
    R = hot ? spline_eval(T.spl[TRX_SPL_R_HOT], M) : spline_eval(T.spl[TRX_SPL_R_COOL], M);
    Te = hot ? spline_eval(T.spl[TRX_SPL_T_HOT], M) : spline_eval(T.spl[TRX_SPL_T_COOL], M);
    if (M != M) { R = 0.0; Te = 0.0; }
    if (R > maxR) R = maxR;
    if (Te > maxT) Te = maxT;
    R = (R < 0.1) ? 0.1 : R;          // NaN caps propagate like torch.clamp_min
    Te = (Te < 2800.0) ? 2800.0 : Te;
}

// funcs.flux_relation (funcs.py:121-140): 10 ** spline(M)
__device__ __forceinline__ double flux_rel(const Tables& T, int which, double M)
{
    return exp(2.302585092994046 * spline_eval(T.spl[which], M));
}

// inverse CDF of a broken power law (priors.py:16-116, 168-383), constants from the host
__device__ __forceinline__ double plaw_inv(const trx_power_law& L, double x)
{
    if (L.ones) return 1.0;
    const double t0 = x / L.norm;
    double arg = 0.0, ip = 0.0;
    bool hit = false;
    for (int j = 0; j < L.nseg; ++j) {       // the last matching segment wins, one power for all lanes
        const bool sel = (j == 0) ? (x <= L.hi[j]) : (x > L.lo[j] && x <= L.hi[j]);
        if (sel) {
            double t = (t0 - L.cum[j]) * L.p1[j];
            if (L.amp[j] != 0.0) t = t / L.amp[j];
            arg = t + L.base[j];
            ip = L.ip[j];
            hit = true;
        }
    }
    return hit ? pow_pos(arg, ip) : x;
}

// np.interp for increasing xp, end values held outside
__device__ __forceinline__ double interp(const double* xp, const double* fp, int n, double x)
{
    if (n == 1) return fp[0];
    if (x != x) return x;
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;               // xp[lo] <= x < xp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    const double w = (x - xp[lo]) / (xp[lo + 1] - xp[lo]);
    return fp[lo] + w * (fp[lo + 1] - fp[lo]);
}

// ln of the bound-companion rate (priors.py:580-984) at |delta_mag| = dm
__device__ __forceinline__ double bound_rate(const trx_draw_args& a, const Tables& T, double dm, bool keep_close, long i)
{
    // (sep_in: the reference's own np.interp of this draw, replayed by the caller -- include/trx.h)
    const double seps = a.dist_pc * (a.sep_in ? a.sep_in[i] : interp(T.cc_con, T.cc_sep, a.n_cc, dm));
    const double s_au = seps * kAu;
    const double lp = log10(sqrt(a.kepler_c * (s_au * s_au * s_au)) / 86400.0);
    const double f1 = a.f1, f2 = a.f2, f3 = a.f3;
    const double alpha = 0.018, dlogP = 0.7;
    const double k = f2 - f1 - alpha * dlogP;
    const double t2p = 0.5 * (lp - 1.0) * (2.0 * f1 + k * (lp - 1.0));
    const double t3p = 0.5 * alpha * (lp * lp - 5.4 * lp + 6.8) + f2 * (lp - 2.0);
    const double k4 = f3 - f2 - alpha * dlogP;
    const double t4p = alpha * dlogP * (lp - 3.4) + f2 * (lp - 3.4) +
                       k4 * (0.238095 * (lp * lp) - 0.952381 * lp + 0.485714);
    const double t5p = f3 * (3.33333 - 17.3566 * exp(-0.3 * lp));
    double f;
    if (keep_close) {
        f = (lp >= 8.0) ? (a.t2 + a.t3 + a.t4 + a.t5)
          : (lp >= 5.5) ? (a.t2 + a.t3 + a.t4 + t5p)
          : (lp >= 3.4) ? (a.t2 + a.t3 + t4p)
          : (lp >= 2.0) ? (a.t2 + t3p)
          : (lp >= 1.0) ? t2p : 0.0;
    } else {
        f = (lp >= 8.0) ? (a.t4 + a.t5) : (lp >= 5.5) ? (a.t4 + t5p) : (lp >= 3.4) ? t4p : 0.0;
    }
    if (lp != lp) f = 0.0;                // torch.where chain: NaN compares false everywhere
    if (a.M_s < 1.0) {
        f = 0.65 * f + 0.35 * f * a.M_s;
        f = (f < 0.0) ? 0.0 : f;
    }
    return log(f);
}

__device__ __forceinline__ double ratio(double f) { return f / (1.0 - f); }

__device__ __forceinline__ double sma(double M_tot, double P_days)
{
    const double ps = P_days * 86400.0;
    return cbrt((kG * M_tot * kMsun) / (4.0 * kPi * kPi) * (ps * ps));
}

// marginal_likelihoods.py:111-123: inc >= inc_min, inc_min = 90 where Ptra > 1 (vector path);
// the per-draw loop skips such draws
__device__ __forceinline__ bool transits(double Ptra, double inc, bool parallel)
{
    const bool ok = Ptra <= 1.0;
    const double c = fmin(fmax(Ptra, -1.0), 1.0);
    const double inc_min = ok ? acos(c) * 180.0 / kPi : 90.0;
    const bool hit = inc >= inc_min;
    return parallel ? hit : (hit && ok);
}

// A draw's uniform random inputs: the staged array, or the kernel's own counter-based stream.  One block yields
// two 53-bit uniforms, and the eight inputs fall into three blocks whose halves no scenario needs both of more
// than once: (R_p | q, inc), (ecc, argp), (q_comp | field index, P).
// (the struct holds the seed, not the argument block: a reference to the kernel's by-value arguments would
// force all 1.2 KB of them into scratch memory)
struct Uniforms {
    const unsigned long long seed;
    const long i;
    double u00, u01, u10, u11, u20, u21;
    bool d0 = false, d1 = false, d2 = false;
    __device__ __forceinline__ Uniforms(unsigned long long seed_, long i_) : seed(seed_), i(i_) {}
    __device__ __forceinline__ double operator()(const double* staged, unsigned slot)
    {
        if (staged) return staged[i];
        if (slot == 2u || slot == 4u || slot == 3u) {
            if (!d0) { philox_uniform2(seed, i, 16u, 0u, u00, u01); d0 = true; }
            return slot == 3u ? u01 : u00;
        }
        if (slot == 5u || slot == 6u) {
            if (!d1) { philox_uniform2(seed, i, 17u, 0u, u10, u11); d1 = true; }
            return slot == 6u ? u11 : u10;
        }
        if (!d2) { philox_uniform2(seed, i, 18u, 0u, u20, u21); d2 = true; }
        return slot == 0u ? u21 : u20;
    }
};

// ---- a cheap necessary condition for the geometry mask -------------------------------------------
// Both masks of a scenario need cos(inc) <= P_tra and P_tra <= 1 (transits(): inc >= acos(P_tra), which no draw
// meets at P_tra > 1), where P_tra = (R_1 + R_2) / a * (1 + e sin w) / (1 - e^2); the twin branch's P_tra at 2 P
// is smaller still.  5-10 % of the draws meet it.  may_transit() evaluates the same chain -- the same
// counter-based random numbers, samplers, mass-radius spline and Kepler's law -- in fp32 on the hardware's
// log2 / exp2 / sin (relative error ~1e-5 at worst) and rejects a draw only when cos(inc) exceeds P_tra by more
// than 0.1 %: the fp64 mask is then evaluated for the draws it lets through (draw_kernel, trx_scenario_enqueue's
// path), regrouped so that its ~1600 fp64 instructions per draw run on full waves of candidates.  NaN anywhere
// compares false and keeps the draw.  tests/test_gpu_fused.py holds the masks with and without the pre-test equal.
__device__ __forceinline__ float pow_f(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

__device__ __forceinline__ float plaw_inv_f(const trx_power_law& L, float x)
{
    if (L.ones) return 1.0f;
    const float t0 = x / (float)L.norm;
    float arg = 0.0f, ip = 0.0f;
    bool hit = false;
    for (int j = 0; j < L.nseg; ++j) {
        const bool sel = (j == 0) ? (x <= (float)L.hi[j]) : (x > (float)L.lo[j] && x <= (float)L.hi[j]);
        if (sel) {
            float t = (t0 - (float)L.cum[j]) * (float)L.p1[j];
            if (L.amp[j] != 0.0) t = t / (float)L.amp[j];
            arg = t + (float)L.base[j];
            ip = (float)L.ip[j];
            hit = true;
        }
    }
    return hit ? pow_f(arg, ip) : x;
}

__device__ __forceinline__ float spline_eval_f(const double* s, float v)
{
    const int m = (int)s[0];
    const double* x = s + 1;
    int i = 0;
    for (int k = 1; k < m; ++k) i += ((float)x[k] <= v) ? 1 : 0;
    const float d = v - (float)x[i];
    const double* c = x + TRX_DRAW_MAX_KNOTS;
    return (((float)c[i] * d + (float)c[TRX_DRAW_MAX_KNOTS + i]) * d + (float)c[2 * TRX_DRAW_MAX_KNOTS + i]) * d +
           (float)c[3 * TRX_DRAW_MAX_KNOTS + i];
}

// radius of funcs.stellar_relations in fp32 (the knots sit within 1e-6 of a node only for one draw in 10^6, and
// there the two branches of the spline agree: it is continuous)
__device__ __forceinline__ float stellar_radius_f(const Tables& T, float M, float maxR)
{
    float R = (M > 0.63f) ? spline_eval_f(T.spl[TRX_SPL_R_HOT], M) : spline_eval_f(T.spl[TRX_SPL_R_COOL], M);
    if (M != M) R = 0.0f;
    if (R > maxR) R = maxR;
    return (R < 0.1f) ? 0.1f : R;
}

__device__ __forceinline__ bool may_transit(const trx_draw_args& a, const Tables& T, const long i)
{
    Uniforms rnd(a.seed, i);
    bool unsure = false;          // an input sits on a branch point of the fp64 chain: left to the fp64 mask
    // a [R_sun] = kSma (M [M_sun] P [d]^2)^(1/3)
    const float kSma = 4.2082785f, kRe = (float)(kRearth / kRsun);
    float P = (float)a.P_lo;
    if (a.uP || a.range_P) P = (float)(a.P_lo + (a.P_hi - a.P_lo) * rnd(a.uP, 0u));
    float mc = 0.0f;
    long k = 0;
    if (a.comp == TRX_COMP_BOUND) {
        const float qc = a.qc_in ? (float)a.qc_in[i] : plaw_inv_f(a.law_qc, (float)rnd(a.uQc, 1u));
        mc = qc * (float)a.M_s;
    } else if (a.comp == TRX_COMP_FIELD) {
        if (a.idx) k = a.idx[i];
        else {
            k = (long)(rnd(nullptr, 7u) * (double)a.n_field_draw);
            k = k < a.n_field_draw ? k : a.n_field_draw - 1;
        }
    }
    float Mh = (float)a.M_s, Rh = (float)a.R_s;
    if (a.host == TRX_HOST_COMPANION) {
        Mh = mc;
        Rh = stellar_radius_f(T, mc, (float)a.R_s);
        unsure = unsure || fabsf(mc - 0.63f) < 1e-4f;          // the two branches of the mass-radius relation
        // The reference looks up the limb-darkening cell of EVERY draw (lnZ_STP / lnZ_SEB: .item() over the unique
        // (Teff, logg) of all N draws) and raises when the Claret grid lacks one, whether that draw transits or not.
        // The cell in fp32; a draw whose cell is missing -- or too close to a rounding boundary of the lattice for
        // fp32 to tell -- goes to the fp64 pass, which raises the flag from the fp64 cell (draw_one).
        {
            float Th = (mc > 0.63f) ? spline_eval_f(T.spl[TRX_SPL_T_HOT], mc) : spline_eval_f(T.spl[TRX_SPL_T_COOL], mc);
            if (mc != mc) Th = 0.0f;
            if (Th > (float)a.Teff) Th = (float)a.Teff;
            Th = (Th < 2800.0f) ? 2800.0f : Th;
            // log10(G M_sun / R_sun^2) = 4.4380676; the hardware log is log2
            const float xg = (4.4380676f + 0.30103f * (__builtin_amdgcn_logf(mc) - 2.0f * __builtin_amdgcn_logf(Rh))) * 2.0f;
            const float xt = Th * (1.0f / 250.0f);
            const bool edge = fabsf(xg - floorf(xg) - 0.5f) < 2e-3f || fabsf(xt - floorf(xt) - 0.5f) < 2e-3f;
            float ig = fminf(fmaxf(rintf(xg) * 0.5f, 3.5f), 5.0f);
            float it = fminf(fmaxf(rintf(xt) * 250.0f, 3500.0f), (float)a.teff_cap);
            float code = rintf((it - 3500.0f) * (1.0f / 250.0f)) * 4.0f + rintf((ig - 3.5f) * 2.0f);
            if (code != code) code = 0.0f;
            int ci = (int)code;
            ci = ci < 0 ? 0 : (ci >= a.n_lut ? a.n_lut - 1 : ci);
            const double cell = T.lut[0][ci];
            unsure = unsure || edge || (cell != cell);
        }
    } else if (a.host == TRX_HOST_FIELD) {
        Mh = (float)a.f_mass[k];
        Rh = (float)a.f_radius[k];
    }
    const float cosi = (float)(1.0 - rnd(a.uInc, 3u));
    const float sinw = __sinf((float)rnd(a.uW, 6u) * 6.2831853f);
    float ecc, size, mtot;
    if (a.planet) {
        if (a.ecc_in) ecc = (float)a.ecc_in[i];
        else {
#if TRX_ECC_ICDF
            ecc = (float)ecc_from_uniform(T, rnd(nullptr, 5u));
#else
            ecc = (float)random_ecc_beta(a.seed, i);
#endif
        }
        const float dRp = (float)rnd(a.uRp, 2u);
        float rp;
        if (a.flat) rp = dRp * 19.5f + 0.5f;
        else rp = (Mh > 0.45f) ? plaw_inv_f(a.law_rp_hi, dRp) : plaw_inv_f(a.law_rp_lo, dRp);
        unsure = unsure || (!a.flat && fabsf(Mh - 0.45f) < 1e-4f);     // the two radius laws
        size = rp * kRe + Rh;
        mtot = Mh;
    } else {
        ecc = pow_f((float)rnd(a.uEcc, 5u), (float)a.ecc_pow);
        const float q = plaw_inv_f(a.law_q, (float)rnd(a.uQ, 4u));
        const float m = q * Mh;
        size = stellar_radius_f(T, m, Rh) + Rh;
        unsure = unsure || fabsf(m - 0.63f) < 1e-4f;
        mtot = Mh + m;
    }
    const float sm = kSma * pow_f(mtot * P * P, 1.0f / 3.0f);
    const float Ptra = size / sm * ((1.0f + ecc * sinw) / (1.0f - ecc * ecc));
    // the twin branch of a binary scenario transits at 2 P: its P_tra is P_tra / 4^(1/3), the only one that can
    // be <= 1 while P_tra itself is not
    const float Pmin = a.planet ? Ptra : Ptra * 0.62996052f;
    const bool reject = (cosi > Ptra * 1.001f + 1e-6f) || (Pmin > 1.001f);
    return unsure || !reject;
}

// One draw: its random inputs, the scenario's parameters, the geometry mask(s), the prior.
//   PHASE 0  everything (trx_draw_scenario: the torch-operator chain reads whole columns)
//   PHASE 1  the mask(s) only -- what does not feed them (flux ratios, the prior) is not computed and no
//            column is written: 90-95 % of the draws fail the geometry and are never looked at again
//   PHASE 2  the columns and the prior of a draw that passed (compact_fill_kernel): the same
//            code on the same counter-based random numbers, hence the same values
//            Where a draw's columns go: PHASE 0 at its own index of the [ncol][N] block (every draw has a place);
//            PHASE 2 where the caller says (col_at, col_stride, prior_at): compact_fill_kernel stores the masked draws
//            DENSELY, in list order -- the likelihood kernels then read coalesced runs instead of one 64-byte line per
//            column and row (the masked draws are one in ten: read in place they cost 700 MB of traffic per target)
template <int PHASE>
__device__ __forceinline__ void draw_one(const trx_draw_args& a, const Tables& T, const long i, const bool parallel,
                                         bool& hit, bool& hit_twin, double* col_at = nullptr, long col_stride = 0,
                                         double* prior_at = nullptr)
{
    const long N = (PHASE == 2) ? col_stride : a.N;
    hit = hit_twin = false;
    Uniforms rnd(a.seed, i);
    double dP = 0.0, dQc = 0.0, dRp = 0.0, dQ = 0.0, dEcc = 0.0, dIdx = 0.0, dBeta = 0.0;
    double P = a.P_lo;
    if (a.uP || a.range_P) { dP = rnd(a.uP, 0u); P = a.P_lo + (a.P_hi - a.P_lo) * dP; }
    // ---- unresolved companion (bound) or field star ----------------------------------
    double frc = 0.0, qc = 1.0, mc = 0.0;
    long k = 0;
    if (a.comp == TRX_COMP_BOUND) {
        if (a.qc_in) qc = a.qc_in[i];
        else { dQc = rnd(a.uQc, 1u); qc = plaw_inv(a.law_qc, dQc); }
        mc = qc * a.M_s;
        const double f = flux_rel(T, TRX_SPL_F_TESS, mc);
        frc = f / (f + a.f0_tess);
    } else if (a.comp == TRX_COMP_FIELD) {
        if (a.idx) k = a.idx[i];
        else {
            dIdx = rnd(nullptr, 7u);
            k = (long)(dIdx * (double)a.n_field_draw);
            k = k < a.n_field_draw ? k : a.n_field_draw - 1;
        }
        frc = a.f_fr[k];
    }
    // ---- host star ---------------------------------------------------------------------
    double Mh = a.M_s, Rh = a.R_s, Th = a.Teff, u1 = a.u1, u2 = a.u2;
    bool extra = true;
    if (a.host == TRX_HOST_COMPANION) {
        Mh = mc;
        stellar_relations(T, mc, a.R_s, a.Teff, Rh, Th);
        const double rm = Rh * kRsun;
        const double logg = log10(kG * (mc * kMsun) / (rm * rm));
        // rounded (Teff / 250 K, logg / 0.5 dex) lattice at the nearest Z
        // (marginal_likelihoods.py:945-972; Teff cap 10000 K for STP, 13000 K for SEB :1181)
        double ig = rint(logg / 0.5) * 0.5;
        ig = fmin(fmax(ig, 3.5), 5.0);
        double it = rint(Th / 250.0) * 250.0;
        it = fmin(fmax(it, 3500.0), a.teff_cap);
        double code = rint((it - 3500.0) / 250.0) * 4.0 + rint((ig - 3.5) / 0.5);
        if (code != code) code = 0.0;
        int ci = (int)code;
        ci = ci < 0 ? 0 : (ci >= a.n_lut ? a.n_lut - 1 : ci);
        u1 = T.lut[0][ci];
        u2 = T.lut[1][ci];
        if (PHASE != 2 && u1 != u1) atomicOr(a.flag, 1);     // a cell the Claret grid lacks: the reference raises
    } else if (a.host == TRX_HOST_FIELD) {
        Mh = a.f_mass[k];
        Rh = a.f_radius[k];
        Th = a.f_teff[k];
        u1 = a.f_u1[k];
        u2 = a.f_u2[k];
        extra = (a.f_logg[k] >= 3.5) && (Th <= 10000.0);      // marginal_likelihoods.py:1950, 2212
    }
    if (a.comp == TRX_COMP_BOUND) extra = extra && (qc != 0.0);

    const double dInc = rnd(a.uInc, 3u), dW = rnd(a.uW, 6u);
    const double inc = acos(1.0 - dInc) * 180.0 / kPi;                  // priors.py:119-132
    const double w = dW * 360.0;                                        // :157-166
    const double sinw = sin(w * kPi / 180.0);
    double ecc, lnprior = 0.0, dm = 0.0;
    bool dm_set = false;
    double* col = (PHASE == 2) ? col_at : a.cols + i;
    if (a.planet) {
        if (a.ecc_in) ecc = a.ecc_in[i];                                // Beta(0.867, 3.030) draws, priors.py:146-148
        else {
#if TRX_ECC_ICDF
            dEcc = rnd(nullptr, 5u);
            ecc = ecc_from_uniform(T, dEcc);
#else
            ecc = random_ecc_beta(a.seed, i);
#endif
        }
        dBeta = ecc;
        dRp = rnd(a.uRp, 2u);
        double rp;
        if (a.flat) rp = dRp * 19.5 + 0.5;
        else rp = (Mh > 0.45) ? plaw_inv(a.law_rp_hi, dRp) : plaw_inv(a.law_rp_lo, dRp);
        const double sm = sma(Mh, P);
        const double size = rp * kRearth + Rh * kRsun;
        const double Ptra = size / sm * ((1.0 + ecc * sinw) / (1.0 - ecc * ecc));
        const bool coll = size > sm * (1.0 - ecc);
        const bool m0 = transits(Ptra, inc, parallel) && !coll && extra;
        hit = m0;
        if (PHASE != 2) a.mask[i] = m0 ? 1 : 0;
        if (PHASE != 1) {
            col[0 * N] = rp; col[1 * N] = P; col[2 * N] = inc; col[3 * N] = sm; col[4 * N] = Rh;
            col[5 * N] = u1; col[6 * N] = u2; col[7 * N] = ecc; col[8 * N] = w; col[9 * N] = frc;
            col[10 * N] = Mh;
        }
        // ---- prior: flux term of the companion / field star ---------------------------
        if (a.prior == TRX_PRIOR_BOUND_TP) {
            double fr = ratio(frc);
            if (a.use_cc) { const double f = flux_rel(T, TRX_SPL_F_BAND, mc); fr = ratio(f / (f + a.f0_band)); }
            dm = 2.5 * log10(fr);
            dm_set = true;
            lnprior = bound_rate(a, T, fabs(dm), false, i);
        }
    } else {
        dEcc = rnd(a.uEcc, 5u);
        ecc = pow_pos(dEcc, a.ecc_pow);                                 // priors.py:146-155
        dQ = rnd(a.uQ, 4u);
        const double q = plaw_inv(a.law_q, dQ);
        const double m = q * Mh;
        double r, tdummy;
        stellar_relations(T, m, Rh, Th, r, tdummy);
        const double fm = flux_rel(T, TRX_SPL_F_TESS, m);
        double fr = fm / (fm + a.f0_tess);
        double fh_band_share = 0.0;
        if (a.host == TRX_HOST_FIELD) {
            // the background star sits at another distance: rescale the pair's flux share
            const double fh = flux_rel(T, TRX_SPL_F_TESS, Mh);
            fr = fr * (frc / (fh / (fh + a.f0_tess)));                  // marginal_likelihoods.py:2147-2159
            if (a.use_cc) { const double fb = flux_rel(T, TRX_SPL_F_BAND, Mh); fh_band_share = fb / (fb + a.f0_band); }
        }
        const double mt = Mh + m;
        const double sm = sma(mt, P), sm2 = sma(mt, 2.0 * P);
        const double e_corr = (1.0 + ecc * sinw) / (1.0 - ecc * ecc);
        const double size = r * kRsun + Rh * kRsun;
        const double Ptra = size / sm * e_corr, Ptra2 = size / sm2 * e_corr;
        const bool coll = size > sm * (1.0 - ecc);
        const bool coll2 = (2.0 * Rh * kRsun) > sm2 * (1.0 - ecc);
        bool m1 = transits(Ptra, inc, parallel) && !coll && (q < 0.95) && extra;
        bool m2 = transits(Ptra2, inc, parallel) && !coll2 && (q >= 0.95) && extra;
        if (!parallel) m2 = m2 && (Ptra <= 1.0);      // the loop `continue`s before the twin test
        hit = m1;
        hit_twin = m2;
        if (PHASE != 2) {
            a.mask[i] = m1 ? 1 : 0;
            a.mask_twin[i] = m2 ? 1 : 0;
        }
        if (PHASE != 1) {
            col[0 * N] = r; col[1 * N] = fr; col[2 * N] = P; col[3 * N] = inc; col[4 * N] = sm;
            col[5 * N] = Rh; col[6 * N] = u1; col[7 * N] = u2; col[8 * N] = ecc; col[9 * N] = w;
            col[10 * N] = frc; col[11 * N] = sm2; col[12 * N] = m; col[13 * N] = Mh;
        }
        if (a.prior == TRX_PRIOR_BOUND_EB) {
            double term = ratio(frc);
            if (a.use_cc) { const double f = flux_rel(T, TRX_SPL_F_BAND, mc); term = ratio(f / (f + a.f0_band)); }
            if (a.host == TRX_HOST_COMPANION) {
                // lnZ_SEB: the companion AND its EB count (marginal_likelihoods.py:1202-1205)
                double fe = fr;
                if (a.use_cc) { const double f = flux_rel(T, TRX_SPL_F_BAND, m); fe = f / (f + a.f0_band); }
                term = term + ratio(fe);
            }
            dm = 2.5 * log10(term);
            dm_set = true;
            lnprior = bound_rate(a, T, fabs(dm), true, i);
        } else if (a.prior == TRX_PRIOR_FIELD && a.host == TRX_HOST_FIELD) {
            // lnZ_BEB: background star + its EB (marginal_likelihoods.py:2161-2208)
            double term;
            if (a.use_cc) {
                const double frc_cc = a.f_frband[k];
                const double fmb = flux_rel(T, TRX_SPL_F_BAND, m);
                const double fr_cc = (fmb / (fmb + a.f0_band)) * (frc_cc / fh_band_share);
                term = ratio(frc_cc) + ratio(fr_cc);
            } else {
                term = ratio(frc) + ratio(fr);
            }
            dm = 2.5 * log10(term);
            dm_set = true;
        }
    }
    if (a.prior == TRX_PRIOR_FIELD) {
        if (!dm_set) {
            // D scenarios and lnZ_BTP: the field star alone
            dm = a.use_cc ? a.f_delta[k] : 2.5 * log10(ratio(frc));
        }
        if (a.use_cc) {
            const double s = a.sep_in ? a.sep_in[i] : interp(T.cc_con, T.cc_sep, a.n_cc, fabs(dm));
            lnprior = log(a.bg_amp * (s * s));
        } else {
            lnprior = a.bg_const;
        }
    }
    if (a.prior == TRX_PRIOR_BOUND_TP || a.prior == TRX_PRIOR_BOUND_EB || a.prior == TRX_PRIOR_FIELD) {
        lnprior = (lnprior > 0.0) ? 0.0 : lnprior;     // clamp_max: NaN stays NaN
        if (dm > 0.0) lnprior = -INFINITY;
    }
    if (PHASE == 0 && a.dm_out) a.dm_out[i] = (dm_set || a.prior == TRX_PRIOR_FIELD) ? dm : NAN;
    if (PHASE == 0 && a.lnprior) a.lnprior[i] = lnprior;
    if (PHASE == 2 && prior_at) *prior_at = lnprior;
    if (PHASE == 0 && a.dump) {
        const long N = a.N;
        double* dd = a.dump + i;
        dd[0 * N] = dP; dd[1 * N] = dQc; dd[2 * N] = dRp; dd[3 * N] = dInc; dd[4 * N] = dQ;
        dd[5 * N] = dEcc; dd[6 * N] = dW; dd[7 * N] = (double)k; dd[8 * N] = dBeta;
    }
}

__device__ __forceinline__ void stage_tables(const trx_draw_args& a, Tables& T)
{
    double* dst = reinterpret_cast<double*>(&T);
    const int nspl = TRX_DRAW_N_SPLINES * TRX_DRAW_SPLINE_DOUBLES;
    for (int i = threadIdx.x; i < nspl; i += blockDim.x) dst[i] = a.splines[i];
    for (int i = threadIdx.x; i < a.n_cc; i += blockDim.x) { T.cc_sep[i] = a.cc_seps[i]; T.cc_con[i] = a.cc_cons[i]; }
    for (int i = threadIdx.x; i < a.n_lut; i += blockDim.x) { T.lut[0][i] = a.lut[i]; T.lut[1][i] = a.lut[a.n_lut + i]; }
    if (a.planet && !a.ecc_in)
        for (int i = threadIdx.x; i <= kEccIcdfN; i += blockDim.x) { T.ecc_lo[i] = kEccIcdfLo[i]; T.ecc_hi[i] = kEccIcdfHi[i]; }
    __syncthreads();
}

// KIND 0 (trx_draw_scenario): every draw in full, grid-stride.
// KIND 1, 2 (trx_scenario_evidence; 2 = with the fp32 pre-test): workgroup b takes the `per` consecutive draws from b * per, writes
// their mask(s) only and leaves the number of its draws that passed in blk_cnt[b] (and blk_cnt[gridDim.x + b]
// for the twin branch) -- the first half of the ordered compaction; its second half and the columns of the draws
// that passed follow in compact_fill_kernel.  A draw's numbers depend on its index only, so the
// mapping of draws to threads changes no result.  (Three kernels, not one with branches: with two inlined
// copies of the draw in one kernel the compiler moved the 1.2 KB argument block into scratch memory.)
template <int KIND>
__device__ __forceinline__ void draw_body(const trx_draw_args& a, int* __restrict__ blk_cnt, long per)
{
    __shared__ Tables T;
    __shared__ int wave_cnt[2][4];
    stage_tables(a, T);
    const long N = a.N;
    const bool parallel = a.parallel != 0;
    int hits = 0, hits_twin = 0;
    if (KIND == 0) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
            bool h0, h1;
            draw_one<0>(a, T, i, parallel, h0, h1);
        }
        return;
    }
    const long i_end = ((long)(blockIdx.x + 1) * per < N) ? (long)(blockIdx.x + 1) * per : N;
    if (KIND == 1) {
        for (long i = (long)blockIdx.x * per + threadIdx.x; i < i_end; i += blockDim.x) {
            bool h0, h1;
            draw_one<1>(a, T, i, parallel, h0, h1);
            hits += h0 ? 1 : 0;
            hits_twin += h1 ? 1 : 0;
        }
    } else {
        // kDrawChunk draws at a time: the fp32 pre-test of all of them (may_transit), the candidates' indices
        // regrouped through LDS, the fp64 mask of the candidates on full lanes
        __shared__ int cand[kDrawChunk];
        __shared__ int ncand;
        for (long c0 = (long)blockIdx.x * per; c0 < i_end; c0 += kDrawChunk) {
            if (threadIdx.x == 0) ncand = 0;
            __syncthreads();
            const long c1 = (c0 + kDrawChunk < i_end) ? c0 + kDrawChunk : i_end;
            for (long i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
                a.mask[i] = 0;
                if (!a.planet) a.mask_twin[i] = 0;
                if (may_transit(a, T, i)) cand[atomicAdd(&ncand, 1)] = (int)(i - c0);
            }
            __syncthreads();
            const int nc = ncand;
            for (int j = threadIdx.x; j < nc; j += blockDim.x) {
                bool h0, h1;
                draw_one<1>(a, T, c0 + cand[j], parallel, h0, h1);
                hits += h0 ? 1 : 0;
                hits_twin += h1 ? 1 : 0;
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hits += __shfl_xor(hits, o, 64);
        hits_twin += __shfl_xor(hits_twin, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { wave_cnt[0][threadIdx.x >> 6] = hits; wave_cnt[1][threadIdx.x >> 6] = hits_twin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = wave_cnt[0][0] + wave_cnt[0][1] + wave_cnt[0][2] + wave_cnt[0][3];
        blk_cnt[gridDim.x + blockIdx.x] = wave_cnt[1][0] + wave_cnt[1][1] + wave_cnt[1][2] + wave_cnt[1][3];
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void draw_kernel(trx_draw_args a, int* __restrict__ blk_cnt, long per)
{
    draw_body<KIND>(a, blk_cnt, per);
}

// One launch chain for several lnZ_* calls (trx_star_enqueue): the call is the grid's second dimension and its
// argument block -- 1.2 KB, too large for several to ride in one kernel's argument buffer -- is read from a table in
// DEVICE memory (uniform addresses: scalar loads, and none of the block is held in registers across the draw).
template <int KIND>
__global__ __launch_bounds__(256) void draw_kernel_star(const trx_draw_args* __restrict__ tab, int* __restrict__ blk_cnt_all, long per)
{
    draw_body<KIND>(tab[blockIdx.y], blk_cnt_all + (long)blockIdx.y * 2 * trx::kDrawMaxGroups, per);
}

// Ordered compaction of the geometry mask(s) AND the columns / prior of the draws that passed, in one kernel
// (compact_kernel + fill_kernel until round 3: two launches, two tails).  grid = (chunks, branches), one wave per
// workgroup.  Workgroup (c, br) owns `gper` consecutive workgroups' worth of draw_kernel's draws, [c gper per,
// (c + 1) gper per): their place in the list idx[br] is the sum of the mask counts of the draw-kernel workgroups
// before them (<= 2048 numbers, summed here: no scan kernel, no cross-workgroup hand-off), their order the draw
// index -- the order numpy's / torch's nonzero gives.  The wave walks its draws 64 at a time, appends the indices
// of the masked ones to idx[br] and to a list in LDS, and whenever that holds a full wave of them (and at the end)
// evaluates draw_one<2> for them: the same code on the same counter-based random numbers as the mask pass, now
// writing the columns and the prior.  A draw passes at most one of the two masks of a binary scenario.  Draw 0 is
// always filled: it stands in for the best draw of a branch no draw passed.
constexpr int kFillList = 128;
#ifndef TRX_FILL_WAVES
#define TRX_FILL_WAVES 4
#endif
// hand-over between the lanes of ONE wave through LDS (its operations are performed in order: a compiler fence is enough)
__device__ __forceinline__ void fill_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// W waves per workgroup, each with draw workgroups of its own (chunk = blockIdx.x * W + wave); they share the staged
// tables (10.5 KB of LDS, staged once per workgroup instead of once per wave) and nothing else.
// Where the masked draws' columns go (dense layout): row r of branch 0 at position r of the [ncol][N] block, row r of the
// twin branch at position N - 1 - r (a draw passes at most one of the two masks, so the two runs never meet); the
// prior likewise.  Draw 0 -- the stand-in for the best draw of a branch no draw passed -- goes to `cols0` [ncol].
template <int W>
__device__ __forceinline__ void compact_fill_body(const trx_draw_args& a, long per, int groups, int gper,
                                                  const int* __restrict__ blk_cnt, int* __restrict__ idx0,
                                                  int* __restrict__ idx1, long* __restrict__ n_out, double* __restrict__ cols0,
                                                  const int n_pad = 1)
{
    __shared__ Tables T;
    __shared__ int hits_all[W][kFillList];
    __shared__ int hpos_all[W][kFillList];
    const int wave = W > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int br = blockIdx.y, lane = W > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    int* hits = hits_all[wave];
    int* hpos = hpos_all[wave];
    const unsigned char* mask = br ? a.mask_twin : a.mask;
    int* idx = br ? idx1 : idx0;
    const int* cnt = blk_cnt + (long)br * groups;
    const int chunk = (int)blockIdx.x * W + wave;
    int g0 = chunk * gper, g1 = (g0 + gper < groups) ? g0 + gper : groups;
    if (g0 > groups) g0 = g1 = groups;                       // (a wave beyond the last chunk: nothing of its own)
    long at = 0;
    for (int j = lane; j < g0; j += 64) at += cnt[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) at += __shfl_xor(at, o, 64);
    int mine = 0;
    for (int j = g0 + lane; j < g1; j += 64) mine += cnt[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (g1 == groups && g0 < groups && lane == 0) n_out[br] = at + mine;
    const bool first = chunk == 0 && br == 0;               // this wave also fills draw 0
    if (W == 1 && mine == 0 && !first) return;
    stage_tables(a, T);                                      // (every thread of the workgroup: it ends with the barrier)
    if (mine == 0 && !first) return;
    const bool parallel = a.parallel != 0;
    const long N = a.N;
    const long end = ((long)g1 * per < N) ? (long)g1 * per : N;
    int nh = 0;
    int pad_next = first ? 0 : n_pad;      // stand-in draws 0 .. n_pad - 1 still to be filled (the first wave's job)
    long i0 = (long)g0 * per;
    // (ONE call site of draw_one: with two inlined copies of the draw in one kernel the compiler once moved the
    // 1.2 KB argument block to scratch memory)
    for (;;) {
        // gather masked draws until a full wave of them is listed or the input is exhausted
        while (nh < 64 && i0 < end) {
            const long i = i0 + lane;
            const bool hit = i < end && mask[i] != 0;
            const unsigned long long m = __ballot(hit);
            i0 += 64;
            if (m == 0) continue;
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (hit) {
                idx[at + below] = (int)i;
                hits[nh + below] = (int)i;
                hpos[nh + below] = (int)(at + below);
            }
            at += __popcll(m);
            nh += __popcll(m);
        }
        if (i0 >= end && pad_next < n_pad) {
            // draw 0 once more, for the stand-in record (position -1: cols0) -- and, when a table of the K best draws is
            // asked for, draws 1 .. K - 1 too (positions -2 ...: the rows of a table that has fewer masked draws than rows)
            int add = n_pad - pad_next;
            add = add < kFillList - nh ? add : kFillList - nh;
            add = add < 64 ? add : 64;
            if (lane < add) { hits[nh + lane] = (int)((long)(pad_next + lane) % N); hpos[nh + lane] = -1 - (pad_next + lane); }
            nh += add;
            pad_next += add;
        }
        if (nh == 0) break;
        fill_wave_sync();
        const int take = nh < 64 ? nh : 64;
        if (lane < take) {
            bool h0, h1;
            const int pos = hpos[lane];
            const long at_col = br ? (N - 1 - (long)pos) : (long)pos;
            double* col_at = (pos < 0) ? cols0 + (-1 - pos) : a.cols + at_col;
            double* prior_at = (pos < 0 || !a.lnprior) ? nullptr : a.lnprior + at_col;
            draw_one<2>(a, T, (long)hits[lane], parallel, h0, h1, col_at, (pos < 0) ? (long)n_pad : N, prior_at);
        }
        fill_wave_sync();
        int carry = 0, carry_pos = 0;
        if (lane < nh - take) { carry = hits[take + lane]; carry_pos = hpos[take + lane]; }
        fill_wave_sync();
        if (lane < nh - take) { hits[lane] = carry; hpos[lane] = carry_pos; }
        nh -= take;
        fill_wave_sync();
    }
}

__global__ __launch_bounds__(64) void compact_fill_kernel(trx_draw_args a, long per, int groups, int gper,
                                                          const int* __restrict__ blk_cnt, int* __restrict__ idx0,
                                                          int* __restrict__ idx1, long* __restrict__ n_out, double* __restrict__ cols0,
                                                          int n_pad)
{
    compact_fill_body<1>(a, per, groups, gper, blk_cnt, idx0, idx1, n_out, cols0, n_pad);
}

// chain: grid = (draw workgroups / kFillWaves, 2 branches, calls); a planet call has one branch and one draw workgroup per wave
struct FillTab {
    trx::ChainFill f[trx::kChainMaxCalls];
};
constexpr int kFillWaves = TRX_FILL_WAVES;
__global__ __launch_bounds__(64 * kFillWaves) void compact_fill_kernel_star(const trx_draw_args* __restrict__ tab, FillTab ft, long per,
                                                                           int groups, const int* __restrict__ blk_cnt_all)
{
    const trx_draw_args& a = tab[blockIdx.z];
    const int gper = a.planet ? 1 : 2;
    if (a.planet && blockIdx.y) return;
    if ((int)blockIdx.x * kFillWaves * gper >= groups) return;
    const trx::ChainFill& f = ft.f[blockIdx.z];
    compact_fill_body<kFillWaves>(a, per, groups, gper, blk_cnt_all + (long)blockIdx.z * 2 * trx::kDrawMaxGroups, f.idx0, f.idx1, f.n_dev,
                                  f.cols0);
}

}  // namespace

extern "C" size_t trx_draw_args_size(void) { return sizeof(trx_draw_args); }

namespace {
int check_draw_args(const trx_draw_args& a)
{
    if (!a.cols || !a.mask || !a.splines || !a.flag) return TRX_ERR_ARG;
    if (!a.planet && !a.mask_twin) return TRX_ERR_ARG;
    if (!a.use_philox) {         // every random input the scenario consumes must be staged
        if (!a.uInc || !a.uW) return TRX_ERR_ARG;
        if (a.planet ? (!a.ecc_in || !a.uRp) : (!a.uEcc || !a.uQ)) return TRX_ERR_ARG;
        if (a.comp == TRX_COMP_BOUND && !a.qc_in && !a.uQc) return TRX_ERR_ARG;
        if ((a.comp == TRX_COMP_FIELD || a.host == TRX_HOST_FIELD) && !a.idx) return TRX_ERR_ARG;
        if (a.range_P && !a.uP) return TRX_ERR_ARG;
    }
    if ((a.comp == TRX_COMP_FIELD || a.host == TRX_HOST_FIELD) && (!a.f_fr || (!a.idx && a.n_field_draw < 1))) return TRX_ERR_ARG;
    if (a.n_cc < 0 || a.n_cc > TRX_DRAW_MAX_CC || a.n_lut < 0 || a.n_lut > TRX_DRAW_MAX_LUT) return TRX_ERR_ARG;
    return TRX_OK;
}
}  // namespace

extern "C" int trx_draw_scenario(const trx_draw_args* args, void* stream)
{
    if (!args || args->N < 0) return TRX_ERR_ARG;
    if (args->N == 0) return TRX_OK;
    const trx_draw_args& a = *args;
    if (int rc = check_draw_args(a)) return rc;
    long blocks = (a.N + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL(draw_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       (int*)nullptr, 0L);
    return hipGetLastError() == hipSuccess ? TRX_OK : TRX_ERR_HIP;
}

// The draw kernel with per-workgroup mask counts (trx_internal.hpp): workgroup b takes draws
// [b * per, (b + 1) * per), blk_cnt holds [2][groups] counts afterwards.
int trx::draw_counted(const trx_draw_args& a, int* blk_cnt, long* per_out, int* groups_out, hipStream_t st)
{
    if (a.N < 1 || !blk_cnt) return TRX_ERR_ARG;
    if (int rc = check_draw_args(a)) return rc;
    long per = (a.N + kDrawMaxGroups - 1) / kDrawMaxGroups;
    per = ((per + 255) / 256) * 256;
    const int groups = (int)((a.N + per - 1) / per);
    if (a.pretest) hipLaunchKernelGGL(draw_kernel<2>, dim3((unsigned)groups), dim3(256), 0, st, a, blk_cnt, per);
    else           hipLaunchKernelGGL(draw_kernel<1>, dim3((unsigned)groups), dim3(256), 0, st, a, blk_cnt, per);
    *per_out = per;
    *groups_out = groups;
    return hipGetLastError() == hipSuccess ? TRX_OK : TRX_ERR_HIP;
}

// Ordered compaction of the mask(s) + the columns and the prior of the draws that passed: see compact_fill_kernel.
// One workgroup of one wave per workgroup of draw_kernel (1024 at N = 1e6: ~1000 draws each, ~100 of which pass at the
// reference's priors -- two waves of fills): 27-39 us for the planet scenarios (profiles/r04/j_draw_kernel.txt; with
// 2048 draw workgroups of 512 draws it was 23-31, and the draw kernel 4 us slower: trx_internal.hpp).
int trx::compact_fill(const trx_draw_args& a, long per, int groups, const int* blk_cnt, int* idx0, int* idx1, long* n_dev,
                      double* cols0, hipStream_t st, int n_pad)
{
    if (a.N < 1 || !idx0 || !n_dev || !blk_cnt || !cols0 || groups < 1 || n_pad < 1) return TRX_ERR_ARG;
    // (binary scenarios: two branches scan the same draws, half as many masked draws each -- two draw workgroups per wave)
    const int gper = a.planet ? 1 : 2;
    const int chunks = (groups + gper - 1) / gper;
    hipLaunchKernelGGL(compact_fill_kernel, dim3((unsigned)chunks, a.planet ? 1u : 2u), dim3(64), 0, st, a, per, groups, gper,
                       blk_cnt, idx0, idx1, n_dev, cols0, n_pad);
    return hipGetLastError() == hipSuccess ? TRX_OK : TRX_ERR_HIP;
}

// The draw kernels of a chain (trx_internal.hpp): masks of every draw of every call, then the ordered lists and the
// columns of the draws that passed -- two launches for all the calls.
int trx::draw_chain(const trx_draw_args* host_args, const trx_draw_args* dev_tab, int n_calls, int* blk_cnt,
                    const ChainFill* fills, long* per_out, int* groups_out, hipStream_t st)
{
    if (!host_args || !dev_tab || !blk_cnt || !fills || n_calls < 1 || n_calls > kChainMaxCalls) return TRX_ERR_ARG;
    const long N = host_args[0].N;
    if (N < 1) return TRX_ERR_ARG;
    bool pretest = true;
    FillTab ft{};
    for (int i = 0; i < n_calls; ++i) {
        if (host_args[i].N != N) return TRX_ERR_ARG;
        if (int rc = check_draw_args(host_args[i])) return rc;
        if (!fills[i].idx0 || !fills[i].n_dev || !fills[i].cols0 || (!host_args[i].planet && !fills[i].idx1)) return TRX_ERR_ARG;
        pretest = pretest && host_args[i].pretest;
        ft.f[i] = fills[i];
    }
    long per = (N + kDrawMaxGroups - 1) / kDrawMaxGroups;
    per = ((per + 255) / 256) * 256;
    const int groups = (int)((N + per - 1) / per);
    if (pretest) hipLaunchKernelGGL(draw_kernel_star<2>, dim3((unsigned)groups, (unsigned)n_calls), dim3(256), 0, st, dev_tab, blk_cnt, per);
    else         hipLaunchKernelGGL(draw_kernel_star<1>, dim3((unsigned)groups, (unsigned)n_calls), dim3(256), 0, st, dev_tab, blk_cnt, per);
    hipLaunchKernelGGL(compact_fill_kernel_star, dim3((unsigned)((groups + kFillWaves - 1) / kFillWaves), 2u, (unsigned)n_calls),
                       dim3(64 * kFillWaves), 0, st, dev_tab, ft, per, groups, (const int*)blk_cnt);
    *per_out = per;
    *groups_out = groups;
    return hipGetLastError() == hipSuccess ? TRX_OK : TRX_ERR_HIP;
}
