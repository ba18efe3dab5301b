/*
 * trx.h -- C ABI of libtrx.so, the MI355X (gfx950) implementation of the
 * marginal-likelihood hot path of stevengiacalone/triceratops.
 *
 * The reference has no FFI: this path sits behind plain in-process Python functions
 * (triceratops/likelihoods.py, triceratops/marginal_likelihoods.py,
 * triceratops/_numerics.py) plus the third-party pytransit.QuadraticModel object.
 * Each entry point below names the reference interface it replaces; INTEGRATION.md
 * shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no C++/torch types; every buffer is caller-owned;
 *   - pointers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()) unless the
 *     function name ends in _host;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream) and the call returns without synchronising: safe to call concurrently on
 *     different streams / devices.  The entry points that evaluate the light-curve model
 *     (trx_lnl_batch, trx_flux_grid, trx_lnz_scenario, trx_scenario_evidence, trx_star_enqueue) keep scratch per
 *     (device, stream) inside the library -- 152 B per row for the per-row constants, the draw
 *     block of trx_scenario_evidence (~0.36 GB per stream at N = 1e6 draws), the arena of a launch chain of
 *     trx_star_enqueue (that much per call of the chain) -- which serves call after call on that stream and only
 *     grows (the stream is synchronised before a buffer is replaced by a larger one);
 *     trx_release_scratch() frees it all.  Two host threads that enqueue on ONE stream take turns (a
 *     per-stream lock is held while a call enqueues its kernels; the stream's order does the rest).
 *     trx_lnl_batch, trx_flux_grid, trx_lnz_scenario and the reductions can be captured
 *     into a hipGraph: while `stream` is capturing, a model call takes a scratch buffer of its own that the
 *     captured graph OWNS (a hipUserObject: it returns to the library's pool when the graph and its executable
 *     graphs are destroyed; rounds 2-5 used graph memory nodes, which intermittently handed a replay zeroed
 *     pages on this stack -- DESIGN.md 4.8).  Inputs must be resident before the capture begins.
 *     NO process-wide mutable state: the library's behaviour is a function of a call's arguments.  What it
 *     keeps besides the scratch are caches whose content cannot change a result -- a mutex-guarded table of the
 *     Gauss nodes per `nsupersample` (filled on first use, read-only afterwards), its device copy, and a
 *     per-light-curve hint whether the centre-value stencil applied (it only saves an empty launch) -- and the
 *     two monotonic row counters of trx_skipped_rows / trx_pruned_rows (statistics).  The tuning and
 *     diagnostics switches that rounds 1-5 exported from this library (trx_set_*) live in include/trx_debug.h
 *     and exist only in the TESTING build, libtrx_testing.so (-DTRX_TESTING); per-call choices are TRX_FLAG_* bits;
 *   - return value: 0 = ok, TRX_ERR_* otherwise (never throws); trx_last_error()
 *     gives a thread-local message for the last non-zero return on this thread;
 *   - numerical exclusions travel in-band exactly as in the reference:
 *     +inf in chi^2/2 = excluded draw, -inf / NaN log-weights = zero weight.
 */
#ifndef TRX_H
#define TRX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model families */
#define TRX_MODEL_TP       0  /* transiting planet:  simulate_TP_transit_p, likelihoods.py:302-358 */
#define TRX_MODEL_EB       1  /* eclipsing binary:   simulate_EB_transit_p, likelihoods.py:361-439;
                                 lnL +inf where secondary depth >= 1.5 sigma, :534-538 */
#define TRX_MODEL_EB_TWIN  2  /* twin EB (q>=0.95 at 2 P_orb): same light curve as EB, no secondary
                                 cut, likelihoods.py:542-587 */
#define TRX_MODEL_RAW      3  /* bare pytransit-shaped evaluate_pv (no unit conversion, no dilution);
                                 trx_flux_grid only */

/* flags (bit-or) */
#define TRX_FLAG_COMPANION_IS_HOST 1 /* `companion_is_host=True`, likelihoods.py:352, 427 */
#define TRX_FLAG_SCALAR_K          2 /* scalar-path radius-ratio rule `abs(k-1)<1e-6`, 1/k for the
                                        secondary (likelihoods.py:121-123, 137) instead of the vector
                                        path's `(k-1)<1e-6` on both (likelihoods.py:406, 418) */

#define TRX_FLAG_FP32_MODEL         4 /* mixed precision (BASELINE config 5): fp64 orbit and geometric
                                        differences, fp32 Mandel-Agol arithmetic, fp64 chi^2 and
                                        log-mean-exp accumulation; ~1e-7 absolute in flux */

#define TRX_FLAG_EVALUATE_EXCLUDED   8 /* trx_lnl_batch / trx_lnz_scenario: evaluate the light curve of a draw that
                                        lnL_EB_p's secondary-eclipse rule excludes anyway (+inf either way): for
                                        benchmarks that count every row.  Default: such a row (half of the EB draws
                                        of a typical run, likelihoods.py:535-538) is not evaluated at all;
                                        trx_skipped_rows counts them */
/* Result-neutral per-call choices (same numbers to rounding; for benchmarks and cross-checks): */
#define TRX_FLAG_ALL_SUBEXPOSURES   16 /* every cell evaluates all `nsupersample` sub-exposures, the reference's literal
                                        algorithm, instead of the 3-9 point Gauss rule of the same discrete measure
                                        where the model is analytic over the exposure (agree to ~1e-13 in flux) */
#define TRX_FLAG_NO_STENCIL         32 /* no centre-value stencil: on a uniform time grid of <= 0.3 exposures per cell
                                        (light curves of 320 points and more) a cell far from every limb contact
                                        otherwise takes its exposure average from the instantaneous flux at the
                                        centres of its 13 nearest cells (one evaluation per cell, error bound 1e-15)
                                        instead of 3-4 Gauss nodes (agree to ~2e-14 in flux) */
#define TRX_FLAG_COUNT_EVALUATIONS  64 /* trx_flux_grid: out_flux receives the NUMBER of model evaluations each cell
                                        cost instead of its flux (the census behind bench.py's roofline line) */
#define TRX_FLAG_FULL_EVALUATION   128 /* trx_scenario_* / trx_star_enqueue: every masked draw is evaluated to the end.
                                        Default: bounded evaluation -- a draw is abandoned once it is shown that it
                                        can neither be the best draw (its chi^2 exceeds the smallest finished one) nor
                                        carry weight in the evidence (its log-weight lies 90 below the largest
                                        finished one; the reduction drops everything 80 below the largest): from its
                                        constants alone when the model can never be as deep as the data, else from its
                                        chi^2 over ~16 probe cells and every cell outside its transit window.  lnZ
                                        agrees to rounding, the best draw is the same, results repeat bit for bit;
                                        light curves of fewer than 48 points are always evaluated in full;
                                        trx_pruned_rows counts the abandoned draws.  The per-row entry points
                                        (trx_lnl_batch, trx_lnz_scenario ...) always return the full chi^2. */

/* parameter-block rows, SoA [n_param][n] contiguous fp64 (reference argument order):
 *   TP  (10): R_p[R_earth] P_orb[d] inc[deg] a[cm] R_s[R_sun] u1 u2 ecc argp[deg] companion_fluxratio
 *   EB  (11): R_EB[R_sun] EB_fluxratio P_orb[d] inc[deg] a[cm] R_s[R_sun] u1 u2 ecc argp[deg] companion_fluxratio
 *   RAW  (9): k t0 p a(in stellar radii) i[rad] e w[rad] u1 u2      (pvp columns + ldc)            */
#define TRX_NPARAM_TP  10
#define TRX_NPARAM_EB  11
#define TRX_NPARAM_RAW 9

/* status codes */
#define TRX_OK              0
#define TRX_ERR_ARG         1  /* NULL pointer, negative size, unknown model ... */
#define TRX_ERR_HIP         2  /* a HIP runtime call failed (see trx_last_error) */
#define TRX_ERR_WORKSPACE   3  /* workspace too small */
#define TRX_ERR_NTOTAL      4  /* N_total != len(logw): the reference raises ValueError,
                                  _numerics.py:40-45 */

/* Replaces lnL_TP_p / lnL_EB_p / lnL_EB_twin_p (likelihoods.py:443-587), i.e. the light-curve
 * model + 0.5*sum((flux-model)^2/sigma^2, axis=1), fused: the (n x n_time) grid is never
 * materialised.  out_halfchi2[n] receives +chi^2/2 (+inf where model==EB and secondary depth
 * >= 1.5*sigma).  Unlike the reference it does not convert the caller's `inc` to radians in
 * place (likelihoods.py:344, 410). */
int trx_lnl_batch(int model, int flags,
                  const double* time, const double* flux, int n_time, double sigma,
                  const double* params, long n,
                  double exptime, int nsupersample,
                  double* out_halfchi2, void* stream);

/* Replaces simulate_TP_transit_p / simulate_EB_transit_p (likelihoods.py:302-439) and, with
 * model == TRX_MODEL_RAW, pytransit's QuadraticModel.set_data + evaluate_pv
 * (likelihoods.py:348-349, 414-415).  out_flux is (n, n_time) row-major; out_secdepth[n] may be
 * NULL (written for EB / EB_TWIN only). */
int trx_flux_grid(int model, int flags,
                  const double* time, int n_time,
                  const double* params, long n,
                  double exptime, int nsupersample,
                  double* out_flux, double* out_secdepth, void* stream);

/* The reference's reduction over a materialised model grid,
 * 0.5*np.sum((flux-model)**2/sigma**2, axis=1) (likelihoods.py:486, 537, 586): HBM-bound row
 * reduction over model_grid (n, n_time). */
int trx_chi2_grid(const double* flux, const double* model_grid, int n_time, long n, double sigma,
                  double* out_halfchi2, void* stream);

/* Bytes of device scratch the reductions below need (a small constant). */
size_t trx_workspace_bytes(void);

/* Replaces _log_mean_exp(logw, N_total=...) (_numerics.py:12-51): log(mean(exp(logw))) with
 * -inf / NaN = zero weight still counted in the denominator, any +inf -> +inf, no finite entry
 * -> -inf.  n_total must equal n (TRX_ERR_NTOTAL otherwise, nothing enqueued).  out: 1 double. */
int trx_log_mean_exp(const double* logw, long n, long n_total, double* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Fused evidence of one scenario branch, the tail of every lnZ_* in marginal_likelihoods.py
 * (e.g. lnZ_TTP 117-154):  lnL_i = -0.5 ln(2 pi) - lnsigma - chi2half_i (+ lnprior_i),
 * lnZ = log( sum_i exp(lnL_i) / n_total ), where the n rows are the draws that passed the
 * geometry mask and the other n_total - n draws have weight zero.
 *   lnprior  [n] or NULL  (lnprior_companion of the P/S/D/B scenarios, already masked)
 *   out_halfchi2 [n]      chi^2/2 per masked draw (needed by the caller for the best-100 table)
 *   out_lnz  1 double */
int trx_lnz_scenario(int model, int flags,
                     const double* time, const double* flux, int n_time, double sigma,
                     const double* params, long n,
                     double exptime, int nsupersample,
                     const double* lnprior, long n_total, double lnsigma,
                     double* out_halfchi2, double* out_lnz,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The tail of trx_lnz_scenario on its own: lnZ from chi^2/2 values already on the device. */
int trx_lnz_from_halfchi2(const double* halfchi2, const double* lnprior, long n, long n_total,
                          double lnsigma, double* out_lnz,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Host-pointer conveniences: stage host buffers to the current device, run the kernels above,
 * copy back and synchronise.  For callers without a device allocator (a ctypes binding on
 * numpy arrays).  They are GPU paths, not CPU fallbacks. */
int trx_lnl_batch_host(int model, int flags,
                       const double* time, const double* flux, int n_time, double sigma,
                       const double* params, long n,
                       double exptime, int nsupersample, double* out_halfchi2);
int trx_flux_grid_host(int model, int flags,
                       const double* time, int n_time,
                       const double* params, long n,
                       double exptime, int nsupersample,
                       double* out_flux, double* out_secdepth);
int trx_log_mean_exp_host(const double* logw, long n, long n_total, double* out);

/* Statistics (monotonic device counters, current device; reading synchronises the device; reset != 0 clears the one
 * read): rows of likelihood calls that were not evaluated because lnL_EB_p's secondary-eclipse rule excludes them
 * anyway (see TRX_FLAG_EVALUATE_EXCLUDED), and draws the bounded evaluation abandoned (see TRX_FLAG_FULL_EVALUATION).
 * They cannot change a result. */
int trx_skipped_rows(unsigned long long* out, int reset);
int trx_pruned_rows(unsigned long long* out, int reset);

/* ------------------------------------------------------------------------------------------
 * The per-draw half of one scenario evidence as ONE kernel (no reference counterpart as a single
 * function: it is the body of every lnZ_* between its np.random draws and its lnL_*_p call,
 * marginal_likelihoods.py:67-123 (TTP) ... 2064-2250 (BEB), with the samplers of priors.py:16-383,
 * the stellar / flux relations of funcs.py:54-140 and the companion priors of priors.py:580-1005).
 * Input: staged random numbers in the reference's draw order (device generator, or numpy's global
 * stream copied over); output: the SoA parameter block trx_lnl_batch takes (all N draws, not yet
 * compacted), the geometry mask(s) and lnprior_companion. */

#define TRX_HOST_TARGET     0   /* T, P, D scenarios: the target star hosts the transit */
#define TRX_HOST_COMPANION  1   /* S scenarios: an unresolved bound companion hosts it */
#define TRX_HOST_FIELD      2   /* B scenarios: a background (TRILEGAL) star hosts it */
#define TRX_COMP_NONE       0
#define TRX_COMP_BOUND      1   /* P, S: bound companion (mass ratio draw or MOLUSC table) */
#define TRX_COMP_FIELD      2   /* D, B: background star drawn from the TRILEGAL population */
#define TRX_PRIOR_NONE      0
#define TRX_PRIOR_BOUND_TP  1   /* lnprior_bound_TP, priors.py:580-782 */
#define TRX_PRIOR_BOUND_EB  2   /* lnprior_bound_EB, priors.py:784-984 */
#define TRX_PRIOR_FIELD     3   /* lnprior_background or its no-contrast-curve constant, :986-1005 */

#define TRX_DRAW_MAX_KNOTS      16
#define TRX_DRAW_SPLINE_DOUBLES (1 + 5 * TRX_DRAW_MAX_KNOTS)  /* m, knots[16], c0..c3[16] */
#define TRX_DRAW_N_SPLINES      6
#define TRX_SPL_R_HOT   0   /* funcs.py:19-51: mass -> radius / Teff, Torres et al. branch */
#define TRX_SPL_T_HOT   1
#define TRX_SPL_R_COOL  2   /*                                         cool-dwarf branch   */
#define TRX_SPL_T_COOL  3
#define TRX_SPL_F_TESS  4   /* funcs.py:81-119: mass -> log10 flux, TESS band */
#define TRX_SPL_F_BAND  5   /*                  ... band of the contrast curve */
#define TRX_DRAW_MAX_CC   256   /* contrast-curve points */
#define TRX_DRAW_MAX_LUT  160   /* (Teff / 250 K) x (logg / 0.5 dex) cells of the companion LDC table */

/* inverse CDF of a broken power law (sample_rp, sample_q, sample_q_companion); the constants are
 * computed on the host in the arithmetic of priors.py: segment j covers lo[j] < x <= hi[j] and maps
 * x to ((x / norm - cum[j]) * p1[j] / amp[j] + base[j]) ** ip[j]   (amp[j] == 0: no division) */
typedef struct {
    int nseg, ones;              /* ones: degenerate law (M_s <= 0.1): every draw is 1 */
    double norm;
    double hi[3], lo[3], cum[3], p1[3], amp[3], base[3], ip[3];
} trx_power_law;

typedef struct {
    long N;
    int planet;                  /* 1: *TP scenario (10-column block), 0: *EB (11 columns + twin) */
    int host, comp, prior;       /* TRX_HOST_*, TRX_COMP_*, TRX_PRIOR_* */
    int parallel;                /* 1: vector-path mask semantics, 0: per-draw-loop semantics */
    int flat;                    /* flatpriors */
    int use_cc;                  /* a contrast curve was given */
    int n_cc, n_lut;
    double P_lo, P_hi;           /* period range (uP != NULL) or the fixed period in P_lo */
    double M_s, R_s, Teff, u1, u2;        /* target star and its limb darkening */
    double ecc_pow;              /* binaries: ecc = uEcc ** ecc_pow (priors.py:152-154) */
    double teff_cap;             /* companion LDC lookup: 10000 (STP) or 13000 (SEB) */
    double f0_tess, f0_band;     /* flux_relation(M_s) in the TESS band / contrast-curve band */
    double dist_pc, kepler_c;    /* 1000 / plx;  4 pi^2 / (G M_ref Msun)        (priors.py:601-640) */
    double f1, f2, f3, t2, t3, t4, t5;     /* Moe & Di Stefano rate constants for M_ref */
    double bg_const, bg_amp;     /* ln((N_comp/0.1)(1/3600)^2 2.2^2);  (N_comp/0.1)(1/3600)^2 */
    trx_power_law law_rp_hi, law_rp_lo, law_q, law_qc;
    /* tables (device) */
    const double* splines;       /* [TRX_DRAW_N_SPLINES][TRX_DRAW_SPLINE_DOUBLES] */
    const double* cc_seps;       /* [n_cc] separations, */
    const double* cc_cons;       /* [n_cc] contrasts (increasing) */
    const double* lut;           /* [2][n_lut] companion limb darkening, NaN = cell absent */
    /* TRILEGAL population (device), indexed by idx */
    const double *f_mass, *f_radius, *f_teff, *f_logg, *f_fr, *f_delta, *f_frband, *f_u1, *f_u2;
    /* staged random numbers (device), [N] each; NULL where the scenario does not draw */
    const double *uP, *uQc, *uRp, *uInc, *uQ, *uEcc, *uW;
    const double* ecc_in;        /* planets: Beta(0.867, 3.03) draws */
    const double* qc_in;         /* MOLUSC mass ratios instead of uQc */
    const long* idx;             /* field-star index per draw */
    /* outputs (device) */
    double* cols;                /* [11][N] planet: R_p P inc a R_s u1 u2 ecc argp comp_fr M_host;
                                    [14][N] binary: R_EB EB_fr P inc a R_s u1 u2 ecc argp comp_fr
                                                    a_twin M_EB M_host */
    unsigned char* mask;         /* [N] geometry mask (binary: the q < 0.95 branch) */
    unsigned char* mask_twin;    /* [N] binary: the q >= 0.95 branch at 2 P_orb */
    double* lnprior;             /* [N] or NULL */
    int* flag;                   /* [1]: bit 0 set when a draw needs a limb-darkening cell the grid
                                    lacks (the reference raises ValueError) */
    /* in-kernel random numbers: with use_philox = 1 every random input left NULL above is generated
     * in the kernel by Philox4x32-10, key = seed, counter = (draw index, slot, sub-draw): slot 0 P,
     * 1 q_companion, 2 R_p, 3 inc, 4 q, 5 ecc (binaries), 6 argp, 7 field-star index (two slots share
     * one counter block: (2 | 4, 3), (5, 6), (1 | 7, 0)).  The planets' Beta(0.867, 3.030) eccentricity is
     * the inverse CDF of the slot-5 uniform (tabulated, fp32, within 1e-6 of the exact quantile).  A draw's
     * numbers depend on (seed, draw index) only, not on N or on the launch geometry. */
    int use_philox;
    int range_P;                 /* the period is drawn from [P_lo, P_hi] (implied by uP != NULL) */
    long n_field_draw;           /* field-star index drawn from [0, n_field_draw) */
    int pretest;                 /* trx_scenario_enqueue only: 1 = the geometry masks of the draws that fail a cheap fp32
                                  * necessary condition are written as 0 without the fp64 evaluation (same masks) */
    unsigned long long seed;
    double* dump;                /* [9][N] or NULL (tests): the random numbers each draw used, rows =
                                    P q_c R_p inc q ecc_u argp index ecc_beta (as uniforms / values) */
    /* Replaying the reference's own interpolation of the contrast curve (funcs.py:222-238: np.interp).  On a curve whose
     * contrasts are NOT monotonic np.interp returns the interval its search ends in, and numpy starts that search from
     * the PREVIOUS draw's interval: the value of draw i depends on draw i - 1.  The kernel bisects (no such memory).  A
     * caller that must reproduce the reference draw for draw -- the seeded validation mode -- asks trx_draw_scenario for
     * every draw's contrast (dm_out [N]: the signed delta magnitude the prior is evaluated at, NaN where the scenario
     * has none), runs np.interp over |dm| itself, in draw order, and hands the separations back (sep_in [N], arcsec):
     * where sep_in is given the kernel takes sep_in[i] instead of interpolating.  Both NULL: the kernel's own search. */
    const double* sep_in;
    double* dm_out;
} trx_draw_args;

int trx_draw_scenario(const trx_draw_args* args, void* stream);
size_t trx_draw_args_size(void);   /* sizeof(trx_draw_args): lets a foreign binding check its layout */

/* One lnZ_* call of calc_probs end to end (the body of e.g. lnZ_TTP, marginal_likelihoods.py:39-172,
 * as calc_probs uses it, triceratops.py:797-817: it keeps the best draw and lnZ of each scenario):
 * draw kernel -> ordered compaction of the geometry mask(s) -> lnL_*_p of the masked draws ->
 * _log_mean_exp(lnL [+ lnprior_companion], N_total = N) -> the draw with the smallest chi^2
 * (first of equals, NaN first: numpy's / torch's argmin).  Planet scenarios give one branch, binary
 * scenarios two (q < 0.95 at P_orb, q >= 0.95 at 2 P_orb).
 *   draw      as for trx_draw_scenario; its output pointers (cols, mask, mask_twin, lnprior, flag,
 *             dump) are ignored: the buffers are stream-ordered scratch of the call
 *   out       HOST, [branches][TRX_SCENARIO_OUT]: the best draw's columns (11 or 14, the layout
 *             of trx_draw_args.cols; draw 0 when no draw passes the mask), then lnZ, then the
 *             number of draws that passed the mask; [16] the number of masked draws that hold the smallest
 *             chi^2 (> 1: the best draw is the FIRST of an exact tie -- numpy's argmin; the reference's
 *             argsort, marginal_likelihoods.py:152, may pick another of them); [17] status: 0, or 1 when a
 *             masked draw's chi^2 was never written by any pass of the likelihood (an internal error:
 *             lnZ is NaN then and the caller must not use the record)
 *   out_flag  HOST, [1]: trx_draw_args.flag
 * Nothing inside the call waits for the device: the masked counts stay there (the likelihood kernels
 * read them from device memory and their grids are sized for a guess), so trx_scenario_evidence is the
 * enqueue below followed by ONE hipStreamSynchronize, after which `out` is filled. */
#define TRX_SCENARIO_OUT 18
typedef struct {
    const trx_draw_args* draw;
    const double* time;          /* [n_time] device */
    const double* flux;          /* [n_time] device */
    int n_time, nsupersample;
    double sigma, lnsigma, exptime;
    int flags;                   /* TRX_FLAG_* of trx_lnl_batch */
    int want_prior;              /* lnprior_companion enters the evidence (P, S, D, B scenarios) */
    double* out;
    int* out_flag;
    /* The reference's table of the best draws (marginal_likelihoods.py:152-171: `(-lnL).argsort()[:100]` and the 14
     * columns gathered at those draws), for callers of lnZ_* themselves (calc_probs reads row 0 only: leave table_rows
     * at 0 or 1).  table_rows = K > 1: `table` receives, per branch b, TRX_TABLE_BRANCH(K) doubles at
     * table + b * TRX_TABLE_BRANCH(K): column c of trx_draw_args.cols at [c * (K + 1), c * (K + 1) + K) -- row j = the
     * masked draw with the j-th smallest chi^2 (NaN last; exact ties: the earlier draw first) --, then at
     * [14 * (K + 1), 15 * (K + 1)) the K + 1 smallest chi^2/2 values themselves (a caller that must reproduce numpy's
     * order among exact ties can see from them whether there are any).  Fewer than K masked draws: the rows beyond
     * them hold draws 0, 1, 2 ... (of all N draws, whatever their mask), as the reference's argsort of equal -inf
     * log-likelihoods may.  Such a call evaluates EVERY masked draw to the end (TRX_FLAG_FULL_EVALUATION is implied:
     * the bounded evaluation leaves only the best draw and the evidence exact) and is enqueued on its own, not in a
     * launch chain.  `table`: pinned host memory (written by the device, valid once the stream has passed the call)
     * or device memory.  K <= TRX_TABLE_MAX_ROWS. */
    int table_rows;
    double* table;
} trx_scenario_args;
#define TRX_TABLE_MAX_ROWS 127
#define TRX_TABLE_BRANCH(K) (15 * ((K) + 1))
int trx_scenario_evidence(const trx_scenario_args* args, void* stream);
/* The same call without the final synchronisation: everything is enqueued on `stream` and the function
 * returns.  args->out / args->out_flag are ignored; the record arrives in
 *   out   HOST, [2 * TRX_SCENARIO_OUT + 1] doubles: the branch records as above, then the flag as a double.
 * Give pinned memory (hipHostMalloc / torch pin_memory) so that the copy is asynchronous; `out` is valid once
 * the stream has passed the call (hipStreamSynchronize, an event).  `args` and everything it points to on the
 * host may be reused as soon as the function returns; the device tables it names must stay alive until the
 * stream has passed the call.  A caller can enqueue every lnZ_* call of a calc_probs on a few streams and wait
 * once. */
int trx_scenario_enqueue(const trx_scenario_args* args, double* out, void* stream);
size_t trx_scenario_args_size(void);

/* The lnZ_* calls of one star of calc_probs (triceratops.py:784-1340: TP, EB, PTP, PEB, STP, SEB, DTP, DEB, BTP,
 * BEB for the target; TP, EB for a nearby star) -- or any other group of calls -- in ONE library call:
 * trx_scenario_enqueue(&calls[i], out[i], streams[i]) for i = 0 .. n_calls - 1, in that order.  A host binding
 * builds the argument blocks of a star once and crosses the FFI once instead of ten times.
 *   calls    HOST, [n_calls]
 *   out      HOST, [n_calls] pointers to pinned records of 2 * TRX_SCENARIO_OUT + 1 doubles
 *   streams  HOST, [n_calls] hipStream_t handles (they may repeat; calls on one stream run in order)
 * Returns the status of the first call that failed (the later ones are not enqueued; *n_done, if given, receives
 * the number of calls that were). */
int trx_star_enqueue(const trx_scenario_args* calls, int n_calls, double* const* out, void* const* streams,
                     int* n_done);
/* Consecutive calls of a trx_star_enqueue that sit on ONE stream and share N, the time stamps (pointer and length),
 * exptime, nsupersample and the precision flag are enqueued as one LAUNCH CHAIN: every kernel of the path once, with
 * the call / branch as a further grid dimension (up to 16 calls or 24 branches per chain; ~11 launches instead of
 * 9-17 per call).  Same records, bit for bit.  The calls of a chain run side by side: `out` of all of them is valid
 * once the stream has passed the call.  (The testing library can switch chains off: trx_debug.h.) */

/* Frees the per-stream scratch described above (every device); all streams must be idle. */
int trx_release_scratch(void);

const char* trx_version(void);
const char* trx_last_error(void);
int trx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* TRX_H */
