// libtrx.so: the host side of the likelihood path and the C ABI (include/trx.h) for gfx950.
//
// The kernels live in two headers that only this file includes:
//   trx_cells.hpp    rowc_kernel (per-row constants, 152 B per row of scratch; the EB rows' secondary-eclipse verdict),
//                    sec_scan_kernel, cells_kernel<MODE, STEP, FP32, LONG, ST, PRUNE> (the light-curve model and its
//                    chi^2: one wavefront per row or per batch of rows, in-window cells filed in LDS, each lane plans a
//                    cell, the (cell, node) pairs dealt to all lanes; centre-value stencil on dense uniform grids; the
//                    passes of the bounded evaluation), pilot_stats_kernel, depth_screen_kernel, and what host and
//                    device share (RowsArgs, batch_rows, batch_plan, set_scratch)
//   trx_reduce.hpp   chi2_grid_kernel (row reduction over a materialised grid) and the log-mean-exp kernels, HBM bound
//   (+ trx_device.hpp: the fp64 device math; trx_draw.hip / trx_scenario.hip: the per-draw kernel and the scenario calls)
// Here: the launch plan (plan_cells), the launchers (launch_cells, lnl_lme_chain), the per-stream scratch and the
// scratch of captured calls, the node tables, the entry points of include/trx.h and -- testing build only -- of
// include/trx_debug.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/trx.h"
#include "trx_device.hpp"
#include "trx_internal.hpp"
#include "trx_knobs.hpp"

namespace {

using namespace trx;

thread_local char g_err[256] = "";

int fail(int code, const char* fmt, const char* a = "", long b = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define TRX_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) return fail(TRX_ERR_HIP, "%s (hip error %ld)", hipGetErrorString(e_), (long)e_); \
    } while (0)

}  // namespace

#include "trx_cells.hpp"
#include "trx_reduce.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// Tuning and diagnostics switches: trx_knobs.hpp.  Compile-time constants in the production library; process-wide
// atomics with setters (include/trx_debug.h) in the testing build, read once per enqueue.

int n_params(int model)
{
    switch (model) {
        case TRX_MODEL_TP: return TRX_NPARAM_TP;
        case TRX_MODEL_EB:
        case TRX_MODEL_EB_TWIN: return TRX_NPARAM_EB;
        case TRX_MODEL_RAW: return TRX_NPARAM_RAW;
        default: return -1;
    }
}

// Node sets for the exposure average: the n-point GAUSS rule of the discrete measure the reference
// averages over (S equally spaced points, weight 1/S each).  It reproduces the S-point average of
// every polynomial of degree <= 2n-1, so for a model that is analytic over the exposure its error
// falls like rho^(-2n) with the distance to the nearest limb contact -- half the nodes of an
// interpolatory rule of the same accuracy.  Nodes = roots of the degree-n orthogonal polynomial
// of the measure (Stieltjes recurrence, roots by bisection between the roots of degree n-1),
// weights = Christoffel numbers; long double, once per launch.  Radii from the measured error
// decay (profiles/r01/q_tier_error.txt): <= ~2e-14 per tier.
bool compute_tiers(TierTable& T, int S)
{
    static const int nn[kTiers] = {3, 4, 5, 6, 7, 8, 9};
    static const double rad[kTiers] = {60.0, 13.0, 6.0, 3.5, 2.7, 2.1, 1.8};
    typedef long double ld;
    static ld xs[4096];               // callers hold fill_tiers' mutex
    const bool usable = S <= 4096;
    for (int s = 1; usable && s <= S; ++s) xs[s - 1] = ((ld)s - 0.5L) / S - 0.5L;
    // recurrence p_{k+1} = (x - al[k]) p_k - be[k] p_{k-1}, norms h[k] = <p_k, p_k>
    ld al[kTierMaxNodes + 1], be[kTierMaxNodes + 1], h[kTierMaxNodes + 1];
    auto poly = [&](int deg, ld x) -> ld {            // monic orthogonal polynomial of degree deg
        ld pm = 0.0L, p = 1.0L;
        for (int k = 0; k < deg; ++k) {
            const ld pn = (x - al[k]) * p - (k > 0 ? be[k] : 0.0L) * pm;
            pm = p;
            p = pn;
        }
        return p;
    };
    const int nmax = nn[kTiers - 1];
    if (usable) {
        for (int k = 0; k <= nmax && k < S; ++k) {
            ld num = 0.0L, den = 0.0L;
            for (int s = 0; s < S; ++s) {
                const ld p = poly(k, xs[s]);
                num += xs[s] * p * p;
                den += p * p;
            }
            h[k] = den / S;
            al[k] = num / den;
            be[k] = (k > 0) ? h[k] / h[k - 1] : h[0];
        }
    }
    bool any = false;
    for (int q = 0; q < kTiers; ++q) {
        const int n = nn[q];
        T.n[q] = 0;
        T.radius[q] = rad[q];
        for (int j = 0; j < kTierMaxNodes; ++j) T.x[q * kTierMaxNodes + j] = T.w[q * kTierMaxNodes + j] = 0.0;
        if (!usable || 2 * n + 2 > S) continue;       // needs degree 2n-1 well below S
        // roots of p_m interlace those of p_{m-1}: build them degree by degree
        ld roots[kTierMaxNodes + 2], prev[kTierMaxNodes + 2];
        int np_ = 0;
        for (int m = 1; m <= n; ++m) {
            ld edges[kTierMaxNodes + 3];
            edges[0] = -0.5L;
            for (int i = 0; i < np_; ++i) edges[i + 1] = prev[i];
            edges[np_ + 1] = 0.5L;
            for (int i = 0; i < m; ++i) {
                ld lo = edges[i], hi = edges[i + 1];
                const bool up = poly(m, hi) > 0.0L;
                for (int it = 0; it < 200; ++it) {
                    const ld mid = 0.5L * (lo + hi);
                    if ((poly(m, mid) > 0.0L) == up) hi = mid; else lo = mid;
                }
                roots[i] = 0.5L * (lo + hi);
            }
            np_ = m;
            for (int i = 0; i < m; ++i) prev[i] = roots[i];
        }
        ld w[kTierMaxNodes], sum = 0.0L;
        for (int j = 0; j < n; ++j) {
            ld acc = 0.0L;
            for (int k = 0; k < n; ++k) {
                const ld p = poly(k, roots[j]);
                acc += p * p / h[k];
            }
            w[j] = 1.0L / acc;
            sum += w[j];
        }
        for (int j = 0; j < n; ++j) {
            T.x[q * kTierMaxNodes + j] = (double)roots[j];
            T.w[q * kTierMaxNodes + j] = (double)(w[j] / sum);
        }
        T.n[q] = n;
        any = true;
    }
    return any;
}

// The table depends on S only and costs ~0.8 ms of long-double work at S = 20 (as much as a small
// kernel): computed once per S and kept (S = 20, the reference's default, and the few other
// values a process ever uses).  The cache is the library's only state besides the diagnostics
// switches; a mutex makes concurrent first calls safe.
bool fill_tiers(TierTable& T, int S)
{
    constexpr int kSlots = 16;
    static std::mutex mu;
    static int keys[kSlots];
    static bool oks[kSlots];
    static TierTable tabs[kSlots];
    static int used = 0, next = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (keys[i] == S) { T = tabs[i]; return oks[i]; }
    const bool ok = compute_tiers(T, S);
    const int slot = (used < kSlots) ? used++ : (next++ % kSlots);
    keys[slot] = S;
    oks[slot] = ok;
    tabs[slot] = T;
    return ok;
}

// The part of the table a kernel takes by value (TierHead) and a device copy of its (node, weight) pairs, one per
// (device, S), uploaded on first use and kept for the life of the process (1.1 KB).  Returns 0, or -1 when HIP fails;
// *usable: S is large enough for at least one tier.
int tier_device(int S, TierHead& head, const double** xw, bool* usable)
{
    TierTable T;
    *usable = fill_tiers(T, S);
    for (int q = 0; q < kTiers; ++q) { head.n[q] = T.n[q]; head.radius[q] = T.radius[q]; }
    static std::mutex mu;
    static std::map<std::pair<int, int>, double*> dev_tabs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    std::lock_guard<std::mutex> lock(mu);
    auto it = dev_tabs.find(std::make_pair(dev, S));
    if (it == dev_tabs.end()) {
        double pairs[2 * kTiers * kTierMaxNodes];
        for (int i = 0; i < kTiers * kTierMaxNodes; ++i) { pairs[2 * i] = T.x[i]; pairs[2 * i + 1] = T.w[i]; }
        // (legal while another stream of this thread is being captured into a graph: nothing here touches that stream)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        double* p = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), sizeof(pairs));
        if (e == hipSuccess) e = hipMemcpy(p, pairs, sizeof(pairs), hipMemcpyHostToDevice);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) { (void)hipGetLastError(); if (p) (void)hipFree(p); return -1; }
        it = dev_tabs.emplace(std::make_pair(dev, S), p).first;
    }
    *xw = it->second;
    return 0;
}

// ---- stencil memo (see cells_kernel) ------------------------------------------------------
struct StencilMemo {
    static constexpr int kSlots = 256;
    struct Key { const void* time; int n_time, S, dev; double exptime; };
    std::mutex mu;
    int* flags = nullptr;            // pinned, host- and device-visible: 0 unknown, 1 no stencil, 2 stencil
    Key keys[kSlots];
    int used = 0, next = 0;
    // the slot of this light curve (a fresh one starts as unknown), or null if pinned memory is not to be had
    int* slot(const Key& k)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!flags) {
            if (hipHostMalloc(reinterpret_cast<void**>(&flags), kSlots * sizeof(int),
                              hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                flags = nullptr;
                return nullptr;
            }
            for (int i = 0; i < kSlots; ++i) flags[i] = 0;
        }
        for (int i = 0; i < used; ++i)
            if (keys[i].time == k.time && keys[i].n_time == k.n_time && keys[i].S == k.S && keys[i].dev == k.dev &&
                keys[i].exptime == k.exptime)
                return flags + i;
        const int i = (used < kSlots) ? used++ : (next++ % kSlots);
        keys[i] = k;
        flags[i] = 0;
        return flags + i;
    }
};
StencilMemo g_stencil_memo;


// what the last launch_cells of this thread did (trx::lnl_draws hands it to the reduction that follows)
thread_local const double* t_last_rowc = nullptr;
thread_local bool t_last_pruned = false;


// the PRUNE instantiations exist for the likelihood mode only
template <int MODE>
void launch_pruned(const RowsArgs& a, hipStream_t st, bool long_rows, bool fp32, unsigned grid, size_t lds)
{
    if constexpr (MODE == MODE_LNL) {
        if (long_rows) {
            if (fp32) hipLaunchKernelGGL((cells_kernel<MODE_LNL, true, true, true, false, true>), dim3(grid), dim3(64), lds, st, a);
            else      hipLaunchKernelGGL((cells_kernel<MODE_LNL, true, false, true, false, true>), dim3(grid), dim3(64), lds, st, a);
        } else {
            if (fp32) hipLaunchKernelGGL((cells_kernel<MODE_LNL, true, true, false, false, true>), dim3(grid), dim3(64 * kBatchWaves), lds, st, a);
            else      hipLaunchKernelGGL((cells_kernel<MODE_LNL, true, false, false, false, true>), dim3(grid), dim3(64 * kBatchWaves), lds, st, a);
        }
    }
}

// What launch_cells and the chain launcher decide before anything is enqueued: the rows per wave and the LDS layout, the
// node tables, whether the bounded evaluation applies, the grids.
struct CellsPlan {
    bool long_rows, prune, split, fp32, step;
    size_t lds;
    unsigned grid_main;        // cells_kernel: the full evaluation, or the passes behind the pilot
    unsigned grid_pilot;
    unsigned grid_rowc;        // rowc_kernel (+ 1 header workgroup)
    unsigned grid_scan;        // sec_scan_kernel<8>
    unsigned grid_screen;      // depth_screen_kernel
    bool passes;               // prune: there may be rows behind the pilot (pilot_stats_kernel and the later passes run)
    int probe_B;               // split: rows per wave of the probe pass (its own LDS layout: probe_lds, probe_wave_doubles)
    size_t probe_lds;
    int probe_wave_doubles;
};

// The probe pass of a split launch (batches of short light curves; bounded evaluation) evaluates its ~16 cells per row with
// the fp32 flux model whatever the call's precision: a row leaves it abandoned -- reporting a lower bound that allows for
// the model's error, fp32_model_slack in trx_cells.hpp -- or on the survivors' list, and the survivors' pass evaluates
// those again from the start in the call's own precision.  No result of the call carries fp32 arithmetic; what changes
// is which rows are abandoned at the margin (a few more survive).
#ifndef TRX_PROBE_FP32
#define TRX_PROBE_FP32 1
#endif
constexpr bool kProbeFp32 = TRX_PROBE_FP32 != 0;

template <int MODE>
int plan_cells(RowsArgs& a, bool long_rows, CellsPlan& P)
{
    // rows per wave (batch_rows); with the row count on the device the LDS layout takes the largest value
    a.forced_B = knob_rows_per_wave();
    a.B = long_rows ? 1 : batch_rows(a.n_dev ? -1 : a.n, a.n_time, a.forced_B);
    a.s2 = a.sigma * a.sigma;
    a.rs2 = 1.0 / a.s2;
    a.dS = (double)a.S;
    a.rS = 1.0 / a.dS;
    a.nbatch = 8 * batch_plan(a.n, a.B).P;          // wave positions of the launch (batch_plan)
    a.debug_bug = knob_debug_bug();
    a.mark_unwritten = (MODE == MODE_LNL) ? 1 : 0;
    bool tiers_ok = false;
    if (tier_device(a.S, a.tiers, &a.tier_xw, &tiers_ok)) return fail(TRX_ERR_HIP, "tier table upload failed%s", "", 0);
    a.use_tiers = tiers_ok && knob_tiers() && !(a.flags & TRX_FLAG_ALL_SUBEXPOSURES);
    // bounded evaluation (trx_scenario_evidence): ~16 probe cells per row; its instantiations carry no stencil
    // (default: light curves of one row per wave only -- measured on calc_probs at N = 1e6: Kepler-10b, 478 points,
    // 46 -> 39 ms; at 100 binned points the batched variant gains or loses ~3 % (the probe phase, the pilot
    // launch and the second window pass eat what the abandoned rows save): profiles/r03/bounded_e2e.txt)
    const int prune_mode = (a.flags & TRX_FLAG_FULL_EVALUATION) ? 0 : (a.prune == 2 ? 2 : knob_bounded());
    // (batched variant: the verdict after the probe phase reads flat + corrections - hrem, and hrem holds only the
    // in-window cells of the window passes done so far -- a valid bound only when the batch's cells fit ONE window;
    // a forced rows-per-wave or a raised trx_set_cell_packing_below can exceed it: those launches evaluate in full)
    const bool prune = MODE == MODE_LNL && a.prune && knob_kepler_stepping() &&
                       (prune_mode == 2 || (prune_mode == 1 && long_rows)) &&
                       (long_rows || (long)a.B * a.n_time <= (long)kCellsWindowBatch) &&
                       a.n_time >= kProbeMinPoints;
    a.prune = prune ? 1 : 0;
    // probe cells per row: every (n_time / kProbeCells)-th stamp (TRX_PROBE_CELLS in the environment: experiments)
    static const int probe_cells = (int)env_long("TRX_PROBE_CELLS", kProbeCells);
    a.pstride = prune ? (a.n_time / probe_cells > 1 ? a.n_time / probe_cells : 2) : 1;
    static const int third_stride = (int)env_long("TRX_THIRD_STRIDE", kThirdStride);
    a.pstride3 = (prune && third_stride > 1 && a.n_time >= 4 * third_stride) ? third_stride : 0;
    a.use_stencil = (!prune && long_rows && a.use_tiers && a.exptime > 0.0 && a.S >= 8 && a.n_time >= 64 &&
                     knob_stencil() && !(a.flags & TRX_FLAG_NO_STENCIL)) ? 1 : 0;
    a.skip_excl = knob_skip_excluded() && !(a.flags & TRX_FLAG_EVALUATE_EXCLUDED);
    a.need_sec = (a.model == TRX_MODEL_EB && MODE == MODE_LNL) ||
                 ((a.model == TRX_MODEL_EB || a.model == TRX_MODEL_EB_TWIN) && a.out_sec != nullptr);
    // LDS: [node tables | atan constants | the staged light curve (short curves)] shared by the workgroup's waves,
    // then per wave [rows, accumulators | pair table | in-window list | cell state (| stencil state)]
    static_assert((kCellsPairs + kCellsWindowLong) % 4 == 0 && (kCellsPairs + kCellsWindowBatch) % 4 == 0 &&
                  sizeof(CellState) % 8 == 0 && sizeof(StencilState) % 8 == 0, "8-byte alignment of the LDS arrays");
    const size_t tables = (size_t)(2 * kTiers * kTierMaxNodes + kAtanRanges * kAtanCols + kTierHeadDoubles) * sizeof(double);
    auto wave_bytes = [&](bool long_variant) -> size_t {
        return (size_t)a.B * (kRowDoubles + 4) * sizeof(double)
             + (kCellsPairs + cells_window(long_variant)) * sizeof(unsigned short) + sizeof(CellState)
             + (long_variant ? sizeof(StencilState) : 0);
    };
    a.tl_off = (int)(tables / sizeof(double));
    size_t shared = tables + (long_rows ? 0 : (size_t)2 * a.n_time * sizeof(double));
    size_t lds = shared + cells_waves(long_rows) * wave_bytes(long_rows);
    if (lds > 64 * 1024) {             // a long curve forced through the batched variant by a test knob
        long_rows = true;
        a.B = 1;
        a.nbatch = a.n;
        shared = tables;
        lds = shared + wave_bytes(true);
    }
    a.wave_off = (int)(shared / sizeof(double));
    a.wave_doubles = (int)(wave_bytes(long_rows) / sizeof(double));
#ifdef TRX_NO_SPLIT
    const bool split = false;              // (A/B builds: batches probe and finish in one kernel, as in round 3)
#else
    const bool split = prune && !long_rows;
#endif
    a.split = split ? 1 : 0;
    // The probe pass of a split launch looks at ~16 cells of a row: with the 3 rows per wave that 200 points allow the
    // other passes, a batch has ~24 in-window probe cells -- one chunk on a third of its lanes, two trips of pairs two
    // thirds full (SQ_THREAD_CYCLES_VALU: 0.48 of the lanes active over the pass, profiles/r05/pmc_chain.sh).  Only the
    // probe cells are filed, so the pass needs no room for a batch's every cell in the in-window list: it takes as many
    // rows per wave as the LDS of five waves per SIMD has room for row blocks (9 at 200 points, 11 at 100), its "window"
    // is the whole batch, and its chunks and trips fill up.
    P.probe_B = a.B;
    P.probe_lds = lds;
    P.probe_wave_doubles = a.wave_doubles;
    a.probe_rows = 0;
    a.wave_floor = a.wave_floor3 = 3200;
    if (split && knob_probe_rows() != 1) {
        const int forced = knob_probe_rows();
        const int per_row = a.n_time / a.pstride + 1;                      // probe cells of a row
        const size_t fixed = (kCellsPairs + cells_window(false)) * sizeof(unsigned short) + sizeof(CellState);
        const size_t room = (32 * 1024 - shared) / cells_waves(false);       // five workgroups per CU
        int B2 = room > fixed ? (int)((room - fixed) / ((kRowDoubles + 4) * sizeof(double))) : a.B;
        if (forced > 1) B2 = forced;
        if (B2 > kCellsMaxRows) B2 = kCellsMaxRows;
        if (B2 * per_row > cells_window(false)) B2 = cells_window(false) / per_row;
        if ((long)B2 * a.n_time > 65535) B2 = 65535 / a.n_time;               // (list entries are 16-bit cell numbers)
        if (B2 > a.B) {
            P.probe_B = B2;
            P.probe_wave_doubles = (int)(((size_t)B2 * (kRowDoubles + 4) * sizeof(double) + fixed) / sizeof(double));
            P.probe_lds = shared + cells_waves(false) * (size_t)P.probe_wave_doubles * sizeof(double);
        }
    }
    P.long_rows = long_rows;
    P.prune = prune;
    P.split = split;
    P.lds = lds;
    P.fp32 = (a.flags & TRX_FLAG_FP32_MODEL) != 0;
    P.step = knob_kepler_stepping() != 0;
    // Row count on the device: `n` is its upper bound (every draw of the scenario), the geometry mask
    // keeps 2-11 % of them, so the grid takes a quarter of the bound -- blocks
    // beyond the batches leave at once, batches beyond the grid are reached by the grid-stride loop.
    // (a workgroup is cells_waves() waves, each with its own batches; a multiple of 8 workgroups: one per XCD)
    const long max_grid = 1L << 20;
    auto grid_for = [&](long batches, bool long_variant, long cap) -> unsigned {
        if (a.n_dev) {
            batches = (batches + 3) / 4;
            batches = batches < 4096 ? 4096 : batches;
        }
        const long groups = (batches + cells_waves(long_variant) - 1) / cells_waves(long_variant);
        long want = 8 * ((groups + 7) / 8);
        if (cap > 0 && want > cap) want = cap;
        return (unsigned)(want < max_grid ? want : max_grid);
    };
    // Workgroups beyond the batches are not free: ~2.4 ns each to dispatch and leave, and while they are dealt out they
    // hold back the workgroups of the other streams' kernels.  A grid sized for the upper bound of a row count that
    // only the device knows is mostly such workgroups (a quarter of the bound: 10 000 of them per launch at N = 10^6,
    // three launches per call; 250 000 one-wave workgroups with one row per wave).  So the grid is capped and the waves
    // stride over the positions beyond it: one workgroup per slot of the chip (1280) for the passes of the bounded
    // evaluation, four per slot for a full evaluation -- which loses nothing against one workgroup per four batches
    // when a launch runs alone (static striding below that does: +6 % at two per slot, +18 % at one) --, four waves
    // per slot with one row per wave.  Measured in one job each (profiles/r04/ab_cap.txt): 64 TOIs on three streams
    // 0.265 -> 0.185 s per step (four streams 0.175), the 75-scenario calc_probs 22.0 -> 20.4 ms, 15 scenarios
    // 3.8 -> 3.1 ms, Kepler-10b 12.0 -> 10.6 ms.  (Environment: experiments only.)
    static const long cap_probe = env_long("TRX_GRID_CAP", 1280);
    static const long cap_plain = env_long("TRX_GRID_CAP_PLAIN", 5120);
    static const long cap_long = env_long("TRX_GRID_CAP_LONG", 16384);
    const long cap = (a.n_dev || prune) ? (long_rows ? cap_long : (prune ? cap_probe : cap_plain)) : 0;
    P.grid_main = grid_for(long_rows ? a.n : a.nbatch, long_rows, cap);
    {
        // pilot rows (evaluated to the end; first values of the running bounds)
        const long np = a.n < kPilotRows ? a.n : kPilotRows;
        const int pilot_B = kPilotB < a.B ? kPilotB : a.B;
        const long pilot_batches = long_rows ? np : (split ? (np + pilot_B - 1) / pilot_B : (np + a.B - 1) / a.B);   // (split: kPilotB pilot rows per wave)
        const long pilot_groups = (pilot_batches + cells_waves(long_rows) - 1) / cells_waves(long_rows);
        P.grid_pilot = (unsigned)(8 * ((pilot_groups + 7) / 8));
        P.passes = a.n_dev || a.n > kPilotRows;
    }
    {
        // (with the row count on the device the grids are guesses, and a workgroup that finds nothing to do still costs
        // its dispatch: a chain's rowc_kernel_star was 70 000 one-wave workgroups for 15 000 blocks of rows -- capped,
        // the workgroups stride: 149 -> 106 us per chain at 512 per branch (1024: 116, 256: 110, 160: 124).  The scan
        // of the open rows is the opposite case: serial chains of ~1000 fp64 instructions, eight rows a workgroup -- 61 us
        // with the guessed grid, 123 at 512 workgroups a branch, 373 at 128 (profiles/r05/trace_env.sh).  Environment:
        // experiments.)
        static const long rowc_cap = env_long("TRX_ROWC_CAP", 512);
        static const long scan_cap = env_long("TRX_SCAN_CAP", 1L << 30);
        long rb = (a.n + 63) / 64;
        if (a.n_dev) rb = (rb + 3) / 4 < 64 ? 64 : (rb + 3) / 4;      // see grid_for; rowc_kernel strides over the rest
        if (a.n_dev && rb > rowc_cap) rb = rowc_cap;
        P.grid_rowc = (unsigned)rb + 1;
        // every row when the depth is asked for; else the open rows, ~3 % of the rows of a likelihood call (which
        // are themselves ~10 % of `n` when that is only the upper bound): workgroups stride over the list
        long sb = (a.n + 63) / 64;
        if (!a.out_sec && a.n_dev) sb = (sb + 3) / 4;
        if (!a.out_sec && a.n_dev && sb > scan_cap) sb = scan_cap;
        P.grid_scan = (unsigned)(sb < 64 ? 64 : (sb > 8192 ? 8192 : sb));
        // the depth screen of the rows behind the pilot, lanes = rows; what it leaves goes to the probe pass
        long sg = (a.n + kScreenRows - 1) / kScreenRows;
        if (a.n_dev) sg = (sg + 3) / 4;
        P.grid_screen = (unsigned)(sg < 8 ? 8 : (sg > 512 ? 512 : sg));
    }
    return TRX_OK;
}

template <int MODE>
int launch_cells(const RowsArgs& a0, hipStream_t st, bool long_rows)
{
    trx::StreamLock turn(st);            // rowc_kernel and cells_kernel of one call share the stream's scratch
    RowsArgs a = a0;
    CellsPlan P;
    if (int rc = plan_cells<MODE>(a, long_rows, P)) return rc;
    long_rows = P.long_rows;
    const bool prune = P.prune, split = P.split, fp32 = P.fp32, step = P.step;
    const size_t lds = P.lds;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    const bool capturing = capture == hipStreamCaptureStatusActive;
    int verdict = 0;                       // of an earlier launch on this light curve: 1 no stencil, 2 stencil
    if (a.use_stencil && !capturing) {     // (a captured launch always carries both instantiations)
        int dev = 0;
        TRX_HIP(hipGetDevice(&dev));
        a.memo = g_stencil_memo.slot({a.time, a.n_time, a.S, dev, a.exptime});
        if (a.memo) verdict = *static_cast<volatile int*>(a.memo);
        if (verdict == 1 || verdict == 2) a.use_stencil = 2;
    }
    // the row constants: 152 B per row of the stream's scratch (+ the flat-model chi^2), filled 64 rows
    // per wave.  While the stream is being captured into a hipGraph the scratch is a buffer of its own that the
    // GRAPH owns (trx::capture_scratch: the stream's buffer must not be grown, nor baked into a graph).
    void* scratch = nullptr;
    const size_t scratch_bytes = launch_scratch_doubles(a.n, split) * sizeof(double);
#ifdef TRX_CAPTURE_GRAPH_MEM
    // (A/B builds only, profiles/r06/graph_stress.py: rounds 2-5 took a pair of graph memory nodes here)
    if (capturing) TRX_HIP(hipMallocAsync(&scratch, scratch_bytes, st));
#else
    if (capturing) TRX_HIP(trx::capture_scratch(st, scratch_bytes, &scratch));
#endif
    else TRX_HIP(trx::stream_scratch(st, 0, scratch_bytes, &scratch));
    set_scratch(a, static_cast<double*>(scratch));
    // From here on a failure may leave the scan's counter non-zero in the stream's scratch (rowc_kernel<true> counts,
    // the cells_kernel behind it resets): cleared on the way out, or the next call's list would start beyond its end
    auto fail_launch = [&](hipError_t e) {
        if (!capturing) (void)hipMemsetAsync(a.scan_count, 0, sizeof(unsigned long long), st);
        return fail(TRX_ERR_HIP, "%s (hip error %ld)", hipGetErrorString(e), (long)e);
    };
    if (a.need_sec) {
        // (the scan's counter is zero in the stream's scratch: cleared at allocation, then by every cells_kernel
        // that follows a scan; a captured call's own buffer holds anything)
        if (capturing) TRX_HIP(hipMemsetAsync(a.scan_count, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(rowc_kernel<true>, dim3(P.grid_rowc), dim3(64), 0, st, a);
        if (a.out_sec) hipLaunchKernelGGL(sec_scan_kernel<64>, dim3(P.grid_scan), dim3(64), 0, st, a);
        else           hipLaunchKernelGGL(sec_scan_kernel<8>, dim3(P.grid_scan), dim3(64), 0, st, a);
    } else {
        hipLaunchKernelGGL(rowc_kernel<false>, dim3(P.grid_rowc), dim3(64), 0, st, a);
    }
    const unsigned g2 = P.grid_main;
    t_last_rowc = a.rowc;
    t_last_pruned = prune;
    if (prune) {
        // pilot rows (evaluated to the end; first values of the running bounds), verdict on probing, the rest
        RowsArgs ap = a;
        ap.part = 1;
        launch_pruned<MODE>(ap, st, long_rows, fp32, P.grid_pilot, lds);
        if (P.passes) {
            hipLaunchKernelGGL(pilot_stats_kernel, dim3(1), dim3(256), 0, st, a.out, a.n, a.n_dev, a.rowc, a.surv_count, a.pstride,
                               a.probe_count);
            if (split) hipLaunchKernelGGL(depth_screen_kernel, dim3(P.grid_screen), dim3(256), 0, st, a);
            ap.part = 2;
            // (the probe pass of a split launch: its rows end abandoned or on the survivors' list -- fp32 flux model, kProbeFp32)
            const bool fp32_probe = fp32 || (kProbeFp32 && split);
            if (split && P.probe_B > a.B) {
                RowsArgs a2 = ap;
                a2.B = P.probe_B; a2.probe_rows = P.probe_B; a2.wave_doubles = P.probe_wave_doubles;
                launch_pruned<MODE>(a2, st, long_rows, fp32_probe, g2, P.probe_lds);
            } else {
                launch_pruned<MODE>(ap, st, long_rows, fp32_probe, g2, lds);
            }
            if (split) {
                // batches of short light curves: the launch above was the probe pass; the rows it left alive, compacted
                // across workgroups, are evaluated to the end here
                ap.part = 3;
                launch_pruned<MODE>(ap, st, long_rows, fp32, g2, lds);
            }
        }
    } else if (long_rows) {
        if (verdict != 2) {
            if (!step)      hipLaunchKernelGGL((cells_kernel<MODE, false, false, true, false>), dim3(g2), dim3(64), lds, st, a);
            else if (fp32)  hipLaunchKernelGGL((cells_kernel<MODE, true, true, true, false>), dim3(g2), dim3(64), lds, st, a);
            else            hipLaunchKernelGGL((cells_kernel<MODE, true, false, true, false>), dim3(g2), dim3(64), lds, st, a);
        }
        if (a.use_stencil && verdict != 1) {
            if (!step)      hipLaunchKernelGGL((cells_kernel<MODE, false, false, true, true>), dim3(g2), dim3(64), lds, st, a);
            else if (fp32)  hipLaunchKernelGGL((cells_kernel<MODE, true, true, true, true>), dim3(g2), dim3(64), lds, st, a);
            else            hipLaunchKernelGGL((cells_kernel<MODE, true, false, true, true>), dim3(g2), dim3(64), lds, st, a);
        }
    } else {
        if (!step)      hipLaunchKernelGGL((cells_kernel<MODE, false, false, false, false>), dim3(g2), dim3(64 * kBatchWaves), lds, st, a);
        else if (fp32)  hipLaunchKernelGGL((cells_kernel<MODE, true, true, false, false>), dim3(g2), dim3(64 * kBatchWaves), lds, st, a);
        else            hipLaunchKernelGGL((cells_kernel<MODE, true, false, false, false>), dim3(g2), dim3(64 * kBatchWaves), lds, st, a);
    }
    const hipError_t launched = hipGetLastError();
#ifdef TRX_CAPTURE_GRAPH_MEM
    if (capturing) TRX_HIP(hipFreeAsync(scratch, st));
#endif
    if (launched != hipSuccess) return fail_launch(launched);
    return TRX_OK;
}

template <int MODE>
int launch_rows(const RowsArgs& a0, hipStream_t st)
{
    const bool batches = a0.n_time > 0 && a0.n_time < knob_cells_below();
    return launch_cells<MODE>(a0, st, !batches);
}

int launch_lme(const double* logw, const double* h, const double* lnprior, double c0, long n,
               long n_total, double* out, void* workspace, size_t workspace_bytes, hipStream_t st)
{
    if (workspace_bytes < trx_workspace_bytes() || !workspace)
        return fail(TRX_ERR_WORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)trx_workspace_bytes());
    const int blocks = lme_blocks(n);
    static_assert(kLmeMaxBlocks == 2048, "lme_blocks");
    double* ws = static_cast<double*>(workspace);
    const uintptr_t al = (uintptr_t)(h ? h : logw) | (uintptr_t)lnprior;
    const int vec_ok = (al % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(lme_partial_kernel<false>, dim3(blocks), dim3(256), 0, st, logw, h, lnprior, c0, n,
                       vec_ok, ws, (const long*)nullptr, (const int*)nullptr, (double*)nullptr, (long*)nullptr,
                       (const double*)nullptr, ScenFinal{});
    TRX_HIP(hipGetLastError());
    hipLaunchKernelGGL(lme_final_kernel, dim3(1), dim3(64), 0, st, ws, blocks, n_total, out);
    TRX_HIP(hipGetLastError());
    return TRX_OK;
}

int check_rows(int model, const double* time, int n_time, const double* params, long n, int S)
{
    if (n_params(model) < 0) return fail(TRX_ERR_ARG, "unknown model%s %ld", "", (long)model);
    if (n < 0 || n_time < 0) return fail(TRX_ERR_ARG, "negative size%s (n=%ld)", "", n);
    if (S < 1) return fail(TRX_ERR_ARG, "nsupersample must be >= 1%s (got %ld)", "", (long)S);
    if (n > 0 && n_time > 0 && (!time || !params)) return fail(TRX_ERR_ARG, "null pointer%s", "", 0);
    return TRX_OK;
}

}  // namespace

// ---- per-stream scratch (trx_internal.hpp) ------------------------------------------------
namespace trx {
namespace {
constexpr int kStageRing = 32;           // pinned blocks per stream for small uploads (pinned_stage_begin)
struct ScratchEntry {
    void* p[kScratchSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[kScratchSlots] = {0, 0, 0, 0};
    std::recursive_mutex mu;             // StreamLock: one call at a time enqueues on the stream
    // ring of pinned staging blocks: a block is reused once the event recorded behind its copy has passed
    void* stage[kStageRing] = {};
    size_t stage_cap[kStageRing] = {};
    hipEvent_t stage_ev[kStageRing] = {};
    bool stage_busy[kStageRing] = {};
    int stage_next = 0;
};
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchEntry> g_scratch;      // (node addresses are stable)

ScratchEntry* scratch_entry(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    return &g_scratch[std::make_pair(dev, st)];
}
}  // namespace

StreamLock::StreamLock(hipStream_t st) : mu_(&scratch_entry(st)->mu) { static_cast<std::recursive_mutex*>(mu_)->lock(); }
StreamLock::~StreamLock() { static_cast<std::recursive_mutex*>(mu_)->unlock(); }

hipError_t stream_scratch(hipStream_t st, int slot, size_t bytes, void** out)
{
    ScratchEntry& en = *scratch_entry(st);
    std::lock_guard<std::recursive_mutex> lock(en.mu);
    hipError_t e = hipSuccess;
    if (en.cap[slot] < bytes) {
        // earlier launches on this stream may still read the old buffer
        if (en.p[slot]) {
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            (void)(slot == 3 ? hipHostFree(en.p[slot]) : hipFree(en.p[slot]));
            en.p[slot] = nullptr;
            en.cap[slot] = 0;
        }
        const size_t want = bytes + bytes / 4 + 256;
        e = (slot == 3) ? hipHostMalloc(&en.p[slot], want, hipHostMallocDefault) : hipMalloc(&en.p[slot], want);
        if (e != hipSuccess) return e;
        en.cap[slot] = want;
        if (slot != 3 && (e = hipMemsetAsync(en.p[slot], 0, kScratchZeroed < want ? kScratchZeroed : want, st)) != hipSuccess)
            return e;
    }
    *out = en.p[slot];
    return hipSuccess;
}

// ---- scratch of a call captured into a hipGraph -----------------------------------------------------------------
// Rounds 2-5 took graph memory nodes (hipMallocAsync / hipFreeAsync on the capturing stream).  On this stack (ROCm 7.2,
// gfx950) a replay enqueued behind work still in flight then intermittently runs cells_kernel on row blocks that read as
// ZERO from a 4-KiB page boundary of the node's allocation to its end -- the rows there come back with the flat-model
// chi^2 (profiles/r06/graph_stress_*.txt: 1 replay in ~2000 without a host synchronisation before it, none with one;
// the once-in-twenty-suite-runs failure of test_entry_points_capture_into_a_hip_graph_and_replay).  The runtime maps a
// node's physical memory when the graph is launched, from the host, whatever the GPU is doing.  So a captured call now
// gets a plain hipMalloc'ed buffer of its own -- allocated under the relaxed capture mode, like the node table of
// tier_device -- that belongs to the graph being captured: a hipUserObject retained by that graph hands the buffer back
// to the pool below when the last graph / executable graph that refers to it is destroyed (the callback may not call
// HIP: it only files the buffer), and a later capture takes a filed buffer that is large enough before allocating.
// trx_release_scratch() frees the filed ones.  If the runtime refuses the user object the buffer simply stays with the
// library until trx_release_scratch().
namespace {
struct CaptureBuf {
    void* p;
    size_t bytes;
    int dev;
};
std::mutex g_capture_mu;
std::vector<CaptureBuf*> g_capture_idle;      // handed back by their graphs (or never owned by one): reusable
std::vector<CaptureBuf*> g_capture_orphans;   // the runtime refused the user object: kept until trx_release_scratch()
long g_capture_live = 0;                      // buffers that a graph still owns
void capture_buf_release(void* ud)
{
    std::lock_guard<std::mutex> lock(g_capture_mu);
    g_capture_idle.push_back(static_cast<CaptureBuf*>(ud));
    --g_capture_live;
}
}  // namespace

hipError_t capture_scratch(hipStream_t st, size_t bytes, void** out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    hipGraph_t graph = nullptr;
    unsigned long long id = 0;
    if ((e = hipStreamGetCaptureInfo_v2(st, &status, &id, &graph, nullptr, nullptr)) != hipSuccess) return e;
    if (status != hipStreamCaptureStatusActive || !graph) return hipErrorStreamCaptureInvalidated;
    CaptureBuf* buf = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_capture_mu);
        size_t best = 0;
        for (size_t i = 0; i < g_capture_idle.size(); ++i) {
            CaptureBuf* c = g_capture_idle[i];
            if (c->dev == dev && c->bytes >= bytes && (!buf || c->bytes < buf->bytes)) { buf = c; best = i; }
        }
        if (buf) g_capture_idle.erase(g_capture_idle.begin() + (long)best);
    }
    if (!buf) {
        // (legal while this thread captures in the global mode: nothing here touches the capturing stream)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        void* p = nullptr;
        e = hipMalloc(&p, bytes);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) return e;
        buf = new CaptureBuf{p, bytes, dev};
    }
    hipUserObject_t obj = nullptr;
    bool owned = false;
    if (hipUserObjectCreate(&obj, buf, capture_buf_release, 1, hipUserObjectNoDestructorSync) == hipSuccess) {
        {
            std::lock_guard<std::mutex> lock(g_capture_mu);
            ++g_capture_live;
        }
        if (hipGraphRetainUserObject(graph, obj, 1, hipGraphUserObjectMove) == hipSuccess) owned = true;
        else { (void)hipGetLastError(); (void)hipUserObjectRelease(obj, 1); }      // (the callback files the buffer)
    } else {
        (void)hipGetLastError();
    }
    if (!owned) {
        // no owner: the buffer must outlive a graph whose lifetime the library cannot see -- it is kept out of the pool
        // (never reused) until trx_release_scratch()
        std::lock_guard<std::mutex> lock(g_capture_mu);
        for (size_t i = 0; i < g_capture_idle.size(); ++i)
            if (g_capture_idle[i] == buf) { g_capture_idle.erase(g_capture_idle.begin() + (long)i); break; }
        g_capture_orphans.push_back(buf);
    }
    *out = buf->p;
    return hipSuccess;
}

void capture_scratch_stats(long* live, long* idle)
{
    std::lock_guard<std::mutex> lock(g_capture_mu);
    *live = g_capture_live;
    *idle = (long)g_capture_idle.size();
}

hipError_t capture_scratch_release_idle()
{
    std::lock_guard<std::mutex> lock(g_capture_mu);
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    for (CaptureBuf* c : g_capture_idle) {
        (void)hipSetDevice(c->dev);
        (void)hipFree(c->p);
        delete c;
    }
    g_capture_idle.clear();
    for (CaptureBuf* c : g_capture_orphans) {
        (void)hipSetDevice(c->dev);
        (void)hipFree(c->p);
        delete c;
    }
    g_capture_orphans.clear();
    return hipSetDevice(cur);
}

int fail_hip(hipError_t e) { return fail(TRX_ERR_HIP, "%s (hip error %ld)", hipGetErrorString(e), (long)e); }

hipError_t pinned_stage_begin(hipStream_t st, size_t bytes, void** block, void** ticket)
{
    ScratchEntry& en = *scratch_entry(st);
    std::lock_guard<std::recursive_mutex> lock(en.mu);
    const int k = en.stage_next;
    en.stage_next = (k + 1) % kStageRing;
    hipError_t e = hipSuccess;
    if (en.stage_busy[k]) {
        // (32 uploads ahead of the device on this stream: wait for the oldest to have been copied)
        if ((e = hipEventSynchronize(en.stage_ev[k])) != hipSuccess) return e;
        en.stage_busy[k] = false;
    }
    if (en.stage_cap[k] < bytes) {
        if (en.stage[k]) (void)hipHostFree(en.stage[k]);
        en.stage[k] = nullptr;
        en.stage_cap[k] = 0;
        const size_t want = bytes + bytes / 2 + 256;
        if ((e = hipHostMalloc(&en.stage[k], want, hipHostMallocDefault)) != hipSuccess) return e;
        en.stage_cap[k] = want;
    }
    if (!en.stage_ev[k] && (e = hipEventCreateWithFlags(&en.stage_ev[k], hipEventDisableTiming)) != hipSuccess) return e;
    *block = en.stage[k];
    *ticket = reinterpret_cast<void*>((intptr_t)(k + 1));
    return hipSuccess;
}

void pinned_stage_end(hipStream_t st, void* ticket)
{
    ScratchEntry& en = *scratch_entry(st);
    std::lock_guard<std::recursive_mutex> lock(en.mu);
    const int k = (int)(intptr_t)ticket - 1;
    if (k < 0 || k >= kStageRing) return;
    if (hipEventRecord(en.stage_ev[k], st) == hipSuccess) en.stage_busy[k] = true;
    else (void)hipGetLastError();
}

bool lnl_chain_applicable(int flags, int n_time, long N, int S)
{
    if (N < 1 || n_time < 1 || S < 1) return false;
    RowsArgs a{};
    a.model = TRX_MODEL_TP; a.flags = flags; a.n_time = n_time; a.sigma = 1.0; a.n = N; a.S = S; a.exptime = 0.0;
    static const long dummy_n = 0;
    a.n_dev = &dummy_n;              // (never read on the host: "the row count lives on the device")
    a.prune = 1;
    const bool batches = n_time < knob_cells_below();
    CellsPlan P;
    if (plan_cells<MODE_LNL>(a, !batches, P) != TRX_OK) return false;
    return P.prune && P.step && P.passes;
}

size_t chain_branch_scratch_bytes(long n_upper) { return launch_scratch_doubles(n_upper, true) * sizeof(double); }

// The likelihood and the reduction of every branch of a chain, one launch per stage with the branch as the grid's
// second dimension (trx_internal.hpp).  Returns kChainNotApplicable -- nothing enqueued -- when the launch would not
// take the bounded evaluation's passes (switched off, fewer than 48 points, a batch beyond one window): the caller
// then enqueues the branches one by one.
int lnl_lme_chain(const ChainBranch* br, int nbr, const double* time, int n_time, long N, double exptime, int S,
                  hipStream_t st)
{
    if (!br || nbr < 1 || nbr > kChainMaxBranches) return fail(TRX_ERR_ARG, "lnl_lme_chain: bad branch count%s %ld", "", (long)nbr);
    if (int rc = check_rows(br[0].model, time, n_time, br[0].cols, N, S)) return rc;
    RowsArgs a{};
    a.model = br[0].model; a.flags = br[0].flags; a.time = time; a.flux = br[0].flux; a.n_time = n_time; a.sigma = br[0].sigma;
    a.params = br[0].cols; a.n = N; a.exptime = exptime; a.S = S; a.out = br[0].h;
    a.n_dev = br[0].n_dev; a.src_idx = br[0].src_idx; a.src_stride = N; a.twin_cols = br[0].twin;
    a.dense = 1;
    a.prune = 1;
    const bool batches = n_time > 0 && n_time < knob_cells_below();
    CellsPlan P;
    if (int rc = plan_cells<MODE_LNL>(a, !batches, P)) return rc;
    if (!P.prune || !P.step || !P.passes) return kChainNotApplicable;
    BranchTab bt{};
    LmeTab lt{};
    bool any_sec = false;
    for (int i = 0; i < nbr; ++i) {
        const ChainBranch& c = br[i];
        if (c.model == TRX_MODEL_RAW || !c.n_dev || !c.src_idx || !c.h || !c.cols || !c.flux || !c.scratch || !c.scan_count ||
            !c.ws || !c.amin_pv || !c.amin_pi || ((uintptr_t)c.h % 16) != 0 ||
            ((c.flags ^ br[0].flags) & kChainSharedFlags))
            return fail(TRX_ERR_ARG, "lnl_lme_chain: bad argument in branch%s %ld", "", (long)i);
        BranchArgs& b = bt.b[i];
        b.model = c.model; b.flags = c.flags; b.twin_cols = c.twin; b.need_sec = (c.model == TRX_MODEL_EB) ? 1 : 0;
        any_sec = any_sec || b.need_sec;
        b.flux = c.flux; b.sigma = c.sigma; b.s2 = c.sigma * c.sigma; b.rs2 = 1.0 / b.s2;
        b.prune_c0 = -0.5 * log(kTwoPi) - c.lnsigma;
        b.params = c.cols; b.out = c.h; b.n_dev = c.n_dev; b.src_idx = c.src_idx; b.prune_lp = c.lnprior;
        b.scratch = c.scratch; b.scan_count = c.scan_count;
        RowsArgs tmp = a;                       // (where set_scratch puts this branch's row blocks and header)
        set_scratch(tmp, c.scratch);
        LmeBranch& l = lt.b[i];
        l.h = c.h; l.lnprior = c.lnprior; l.c0 = b.prune_c0; l.ws = c.ws; l.n_dev = c.n_dev;
        l.amin_pv = c.amin_pv; l.amin_pi = c.amin_pi; l.bounds_base = tmp.rowc; l.fin = *c.fin;
    }
    const unsigned y = (unsigned)nbr;
    hipLaunchKernelGGL(rowc_kernel_star, dim3(P.grid_rowc, y), dim3(64), 0, st, a, bt);
    if (any_sec) {
        BranchMap map{};
        unsigned n_sec = 0;
        for (int i = 0; i < nbr; ++i)
            if (bt.b[i].need_sec) map.id[n_sec++] = (unsigned char)i;
        hipLaunchKernelGGL(sec_scan_kernel_star, dim3(P.grid_scan, n_sec), dim3(64), 0, st, a, bt, map);
    }
    // (the few-rows rule of the listed passes asks for ~3200 waves a launch: a chain's launch is all its branches'.  The
    // survivors' pass of a 64-target step ran one row per wave in most branches under the per-launch rule -- lanes 0.65
    // active: same job, 0.114 -> 0.107-0.109 s per step with the floor divided; any value from 1 to 300 does the same,
    // profiles/r05/ab_probe_rows.txt.  Environment: experiments.)
    static const int floor_env = (int)env_long("TRX_WAVE_FLOOR", 0);
    static const int floor3_env = (int)env_long("TRX_WAVE_FLOOR3", 0);
    a.wave_floor = floor_env > 0 ? floor_env : (3200 / nbr > 256 ? 3200 / nbr : 256);
    a.wave_floor3 = floor3_env > 0 ? floor3_env : a.wave_floor;
    auto cells = [&](unsigned grid, int part) {
        if (P.long_rows) {
            if (P.fp32) hipLaunchKernelGGL((cells_kernel_star<true, true>), dim3(grid, y), dim3(64), P.lds, st, a, bt, part);
            else        hipLaunchKernelGGL((cells_kernel_star<false, true>), dim3(grid, y), dim3(64), P.lds, st, a, bt, part);
        } else {
            // (the probe pass of a split launch: more rows per wave, its own LDS layout -- plan_cells)
            RowsArgs ax = a;
            size_t lds = P.lds;
            if (part == 2 && P.split && P.probe_B > a.B) {
                ax.B = P.probe_B; ax.probe_rows = P.probe_B; ax.wave_doubles = P.probe_wave_doubles;
                lds = P.probe_lds;
            }
            if (P.fp32 || (kProbeFp32 && part == 2 && P.split))
                        hipLaunchKernelGGL((cells_kernel_star<true, false>), dim3(grid, y), dim3(64 * kBatchWaves), lds, st, ax, bt, part);
            else        hipLaunchKernelGGL((cells_kernel_star<false, false>), dim3(grid, y), dim3(64 * kBatchWaves), lds, st, ax, bt, part);
        }
    };
    cells(P.grid_pilot, 1);
    hipLaunchKernelGGL(pilot_stats_kernel_star, dim3(1, y), dim3(256), 0, st, a, bt);
    if (P.split) hipLaunchKernelGGL(depth_screen_kernel_star, dim3(P.grid_screen, y), dim3(256), 0, st, a, bt);
    cells(P.grid_main, 2);
    if (P.split) cells(P.grid_main, 3);
    hipLaunchKernelGGL(lme_partial_kernel_star, dim3((unsigned)lme_blocks(N), y), dim3(256), 0, st, lt, N);
    TRX_HIP(hipGetLastError());
    return TRX_OK;
}

int lnl_draws(int model, int flags, const double* time, const double* flux, int n_time, double sigma,
              const double* cols, long n_upper, const long* n_dev, const int* src_idx, long src_stride,
              int twin, double exptime, int nsupersample, double* out_halfchi2, const double* lnprior,
              double lnsigma, const double** bounds_base, hipStream_t st)
{
    *bounds_base = nullptr;
    if (int rc = check_rows(model, time, n_time, cols, n_upper, nsupersample)) return rc;
    if (model == TRX_MODEL_RAW || !n_dev || !src_idx || !out_halfchi2 || n_upper < 1 || (n_time > 0 && !flux))
        return fail(TRX_ERR_ARG, "lnl_draws: bad argument%s", "", 0);
    RowsArgs a{};
    a.model = model; a.flags = flags; a.time = time; a.flux = flux; a.n_time = n_time; a.sigma = sigma;
    a.params = cols; a.n = n_upper; a.exptime = exptime; a.S = nsupersample; a.out = out_halfchi2;
    a.n_dev = n_dev; a.src_idx = src_idx; a.src_stride = src_stride; a.twin_cols = twin;
    a.dense = 1;              // compact_fill_kernel stored the masked draws' columns and priors densely, in list order
    // the caller keeps only lnZ and the best draw of these rows: bounded evaluation (cells_body, PRUNE)
    a.prune = 1;
    a.prune_c0 = -0.5 * log(kTwoPi) - lnsigma;
    a.prune_lp = lnprior;
    const int rc = launch_rows<MODE_LNL>(a, st);
    if (rc == TRX_OK && t_last_pruned) *bounds_base = t_last_rowc;
    return rc;
}

int lme_draws(const double* halfchi2, const double* lnprior, double lnsigma, long n_upper, const long* n_dev,
              const int* src_idx, double* ws, double* amin_pv, long* amin_pi, const double* bounds_base,
              const ScenFinal& fin, hipStream_t st)
{
    if (!halfchi2 || !n_dev || !src_idx || !ws || !amin_pv || !amin_pi || ((uintptr_t)halfchi2 % 16) != 0)
        return fail(TRX_ERR_ARG, "lme_draws: bad argument%s", "", 0);
    const double c0 = -0.5 * log(kTwoPi) - lnsigma;   // marginal_likelihoods.py:130 etc.
    hipLaunchKernelGGL(lme_partial_kernel<true>, dim3(lme_blocks(n_upper)), dim3(256), 0, st, (const double*)nullptr,
                       halfchi2, lnprior, c0, n_upper, 1, ws, n_dev, src_idx, amin_pv, amin_pi, bounds_base, fin);
    TRX_HIP(hipGetLastError());
    return TRX_OK;
}
}  // namespace trx

// =========================================================================================
extern "C" {

int trx_lnl_batch(int model, int flags, const double* time, const double* flux, int n_time,
                  double sigma, const double* params, long n, double exptime, int nsupersample,
                  double* out_halfchi2, void* stream)
{
    if (int rc = check_rows(model, time, n_time, params, n, nsupersample)) return rc;
    if (model == TRX_MODEL_RAW) return fail(TRX_ERR_ARG, "TRX_MODEL_RAW has no likelihood%s", "", 0);
    if (n == 0) return TRX_OK;
    if (!out_halfchi2 || (n_time > 0 && !flux)) return fail(TRX_ERR_ARG, "null pointer%s", "", 0);
    RowsArgs a{};
    a.model = model; a.flags = flags; a.time = time; a.flux = flux; a.n_time = n_time; a.sigma = sigma;
    a.params = params; a.n = n; a.exptime = exptime; a.S = nsupersample; a.out = out_halfchi2;
    if (knob_bounded_lnl()) { a.prune = 2; a.prune_c0 = 0.0; a.prune_lp = nullptr; }
    return launch_rows<MODE_LNL>(a, static_cast<hipStream_t>(stream));
}

int trx_flux_grid(int model, int flags, const double* time, int n_time, const double* params,
                  long n, double exptime, int nsupersample, double* out_flux, double* out_secdepth,
                  void* stream)
{
    if (int rc = check_rows(model, time, n_time, params, n, nsupersample)) return rc;
    if (n == 0) return TRX_OK;
    if (n_time > 0 && !out_flux) return fail(TRX_ERR_ARG, "null pointer%s", "", 0);
    RowsArgs a{};
    a.model = model; a.flags = flags; a.time = time; a.n_time = n_time; a.sigma = 1.0;
    a.params = params; a.n = n; a.exptime = exptime; a.S = nsupersample; a.out = out_flux; a.out_sec = out_secdepth;
    a.debug_nodes = (knob_debug_nodes() || (flags & TRX_FLAG_COUNT_EVALUATIONS)) ? 1 : 0;
    return launch_rows<MODE_GRID>(a, static_cast<hipStream_t>(stream));
}

int trx_chi2_grid(const double* flux, const double* model_grid, int n_time, long n, double sigma,
                  double* out_halfchi2, void* stream)
{
    if (n < 0 || n_time < 0) return fail(TRX_ERR_ARG, "negative size%s (n=%ld)", "", n);
    if (n == 0) return TRX_OK;
    if (!out_halfchi2 || (n_time > 0 && (!flux || !model_grid))) return fail(TRX_ERR_ARG, "null pointer%s", "", 0);
    const int vec_ok = (n_time % 2 == 0) && (((uintptr_t)flux | (uintptr_t)model_grid) % 16 == 0);
    long blocks = (n + 3) / 4;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL(chi2_grid_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), flux, model_grid, n_time, n, sigma,
                       out_halfchi2, vec_ok);
    TRX_HIP(hipGetLastError());
    return TRX_OK;
}

size_t trx_workspace_bytes(void) { return (size_t)kLmeMaxBlocks * 3 * sizeof(double); }

int trx_log_mean_exp(const double* logw, long n, long n_total, double* out, void* workspace,
                     size_t workspace_bytes, void* stream)
{
    if (n < 0 || !out || (n > 0 && !logw)) return fail(TRX_ERR_ARG, "bad argument%s", "", 0);
    if (n_total != n)
        return fail(TRX_ERR_NTOTAL, "N_total must equal len(logw)%s (len=%ld)", "", n);
    return launch_lme(logw, nullptr, nullptr, 0.0, n, n_total, out, workspace, workspace_bytes,
                      static_cast<hipStream_t>(stream));
}

int trx_lnz_from_halfchi2(const double* halfchi2, const double* lnprior, long n, long n_total,
                          double lnsigma, double* out_lnz, void* workspace,
                          size_t workspace_bytes, void* stream)
{
    if (!out_lnz || n < 0 || (n > 0 && !halfchi2)) return fail(TRX_ERR_ARG, "bad argument%s", "", 0);
    if (n_total < n || n_total < 1) return fail(TRX_ERR_NTOTAL, "n_total must be >= n%s (n=%ld)", "", n);
    const double c0 = -0.5 * log(kTwoPi) - lnsigma;   // marginal_likelihoods.py:130 etc.
    return launch_lme(nullptr, halfchi2, lnprior, c0, n, n_total, out_lnz, workspace,
                      workspace_bytes, static_cast<hipStream_t>(stream));
}

int trx_lnz_scenario(int model, int flags, const double* time, const double* flux, int n_time,
                     double sigma, const double* params, long n, double exptime, int nsupersample,
                     const double* lnprior, long n_total, double lnsigma, double* out_halfchi2,
                     double* out_lnz, void* workspace, size_t workspace_bytes, void* stream)
{
    if (int rc = trx_lnl_batch(model, flags, time, flux, n_time, sigma, params, n, exptime,
                               nsupersample, out_halfchi2, stream))
        return rc;
    return trx_lnz_from_halfchi2(out_halfchi2, lnprior, n, n_total, lnsigma, out_lnz, workspace,
                                 workspace_bytes, stream);
}

// ---- host-pointer conveniences -----------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    double* d() { return static_cast<double*>(p); }
};
}  // namespace

int trx_lnl_batch_host(int model, int flags, const double* time, const double* flux, int n_time,
                       double sigma, const double* params, long n, double exptime,
                       int nsupersample, double* out_halfchi2)
{
    if (int rc = check_rows(model, time, n_time, params, n, nsupersample)) return rc;
    if (n == 0) return TRX_OK;
    if (model == TRX_MODEL_RAW) return fail(TRX_ERR_ARG, "TRX_MODEL_RAW has no likelihood%s", "", 0);
    const int np = n_params(model);
    DevBuf dt, df, dp, dout;
    TRX_HIP(dt.alloc(sizeof(double) * n_time));
    TRX_HIP(df.alloc(sizeof(double) * n_time));
    TRX_HIP(dp.alloc(sizeof(double) * np * n));
    TRX_HIP(dout.alloc(sizeof(double) * n));
    TRX_HIP(hipMemcpy(dt.p, time, sizeof(double) * n_time, hipMemcpyHostToDevice));
    TRX_HIP(hipMemcpy(df.p, flux, sizeof(double) * n_time, hipMemcpyHostToDevice));
    TRX_HIP(hipMemcpy(dp.p, params, sizeof(double) * np * n, hipMemcpyHostToDevice));
    if (int rc = trx_lnl_batch(model, flags, dt.d(), df.d(), n_time, sigma, dp.d(), n, exptime,
                               nsupersample, dout.d(), nullptr))
        return rc;
    TRX_HIP(hipMemcpy(out_halfchi2, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TRX_OK;
}

int trx_flux_grid_host(int model, int flags, const double* time, int n_time, const double* params,
                       long n, double exptime, int nsupersample, double* out_flux,
                       double* out_secdepth)
{
    if (int rc = check_rows(model, time, n_time, params, n, nsupersample)) return rc;
    if (n == 0) return TRX_OK;
    const int np = n_params(model);
    DevBuf dt, dp, dout, dsec;
    TRX_HIP(dt.alloc(sizeof(double) * n_time));
    TRX_HIP(dp.alloc(sizeof(double) * np * n));
    TRX_HIP(dout.alloc(sizeof(double) * n * n_time));
    TRX_HIP(dsec.alloc(sizeof(double) * n));
    TRX_HIP(hipMemcpy(dt.p, time, sizeof(double) * n_time, hipMemcpyHostToDevice));
    TRX_HIP(hipMemcpy(dp.p, params, sizeof(double) * np * n, hipMemcpyHostToDevice));
    TRX_HIP(hipMemset(dsec.p, 0, sizeof(double) * n));
    if (int rc = trx_flux_grid(model, flags, dt.d(), n_time, dp.d(), n, exptime, nsupersample,
                               dout.d(), dsec.d(), nullptr))
        return rc;
    TRX_HIP(hipMemcpy(out_flux, dout.p, sizeof(double) * n * n_time, hipMemcpyDeviceToHost));
    if (out_secdepth) TRX_HIP(hipMemcpy(out_secdepth, dsec.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TRX_OK;
}

int trx_log_mean_exp_host(const double* logw, long n, long n_total, double* out)
{
    if (n < 0 || !out || (n > 0 && !logw)) return fail(TRX_ERR_ARG, "bad argument%s", "", 0);
    if (n_total != n)
        return fail(TRX_ERR_NTOTAL, "N_total must equal len(logw)%s (len=%ld)", "", n);
    DevBuf dx, dws, dout;
    TRX_HIP(dx.alloc(sizeof(double) * n));
    TRX_HIP(dws.alloc(trx_workspace_bytes()));
    TRX_HIP(dout.alloc(sizeof(double)));
    TRX_HIP(hipMemcpy(dx.p, logw, sizeof(double) * n, hipMemcpyHostToDevice));
    if (int rc = trx_log_mean_exp(dx.d(), n, n_total, dout.d(), dws.p, trx_workspace_bytes(), nullptr))
        return rc;
    TRX_HIP(hipMemcpy(out, dout.p, sizeof(double), hipMemcpyDeviceToHost));
    return TRX_OK;
}

/* statistics (include/trx.h): rows skipped on the current device since the last reset */
static int read_row_stat(int which, unsigned long long* out, int reset)
{
    static unsigned long long host[kStatShards][kStatPad];
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    constexpr size_t slice = sizeof(host);                 // one counter's shards
    TRX_HIP(hipDeviceSynchronize());
    TRX_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_row_stats), slice, (size_t)which * slice));
    if (out) {
        unsigned long long sum = 0;
        for (int i = 0; i < kStatShards; ++i) sum += host[i][0];
        *out = sum;
    }
    if (reset) {
        // only this counter's slice is cleared (a copy of the WHOLE table written back would roll back what kernels of
        // another host thread added to the other counter between the read and the write: advisor, round 4)
        memset(host, 0, slice);
        TRX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_row_stats), host, slice, (size_t)which * slice));
    }
    return TRX_OK;
}

int trx_skipped_rows(unsigned long long* out, int reset) { return read_row_stat(0, out, reset); }

/* statistics (include/trx.h): rows abandoned by the bounded evaluation on the current device since the last reset */
int trx_pruned_rows(unsigned long long* out, int reset) { return read_row_stat(1, out, reset); }

#ifdef TRX_TESTING
// ---- include/trx_debug.h: the testing library only ---------------------------------------------------------
// (tests, host only: no device is touched) the plan the batched cells_kernel deals its rows by -- walks every position
// of the eight XCDs exactly as cells_body does and checks that the batches tile [0, rows) once, in order within an
// XCD, none larger than rows_per_wave; *positions = positions per XCD, *waves_min = the smallest batch met
int trx_debug_batch_plan(long rows, int rows_per_wave, int taper, long* positions, int* rows_min)
{
    if (rows < 0 || rows_per_wave < 1 || rows_per_wave > kCellsMaxRows) return fail(TRX_ERR_ARG, "bad plan request%s (%ld rows)", "", rows);
    const BatchPlan p = batch_plan(rows, rows_per_wave, taper != 0);
    long covered = 0;
    int smallest = rows_per_wave;
    for (long xcd = 0; xcd < 8; ++xcd) {
        long expect = xcd * p.R;                         // an XCD's batches follow one another in its share
        for (long pos = 0; pos < p.P; ++pos) {
            long base;
            int nb;
            batch_at(p, xcd, pos, base, nb);
            if (base >= rows) continue;
            if (rows - base < nb) nb = (int)(rows - base);
            if (nb < 1 || nb > rows_per_wave || base != expect)
                return fail(TRX_ERR_ARG, "batch plan broken%s (at row %ld)", "", base);
            expect = base + nb;
            covered += nb;
            if (nb < smallest) smallest = nb;
        }
        const long share_end = (xcd + 1) * p.R < rows ? (xcd + 1) * p.R : rows;
        if (xcd * p.R < rows && expect != share_end) return fail(TRX_ERR_ARG, "batch plan leaves rows out%s (after row %ld)", "", expect);
    }
    if (covered != rows) return fail(TRX_ERR_ARG, "batch plan covers the wrong number of rows%s (%ld)", "", covered);
    if (positions) *positions = p.P;
    if (rows_min) *rows_min = smallest;
    return TRX_OK;
}

int trx_set_rows_per_wave(int rows)
{
    if (rows < 0 || rows > kCellsMaxRows)
        return fail(TRX_ERR_ARG, "rows per wave must be 0 (automatic) .. 22%s (got %ld)", "", (long)rows);
    g_knob_rows_per_wave = rows;
    return TRX_OK;
}

/* diagnostics (include/trx.h): light curves with fewer points than this are processed in batches of rows per wave */
int trx_set_cell_packing_below(int n_time)
{
    if (n_time < 0) return fail(TRX_ERR_ARG, "n_time threshold must be >= 0%s (got %ld)", "", (long)n_time);
    g_knob_cells_below = n_time;
    return TRX_OK;
}

/* diagnostics (include/trx.h): 0 = evaluate the light curve of every row, also of those the EB rule excludes */
int trx_set_skip_excluded(int on)
{
    g_knob_skip_excluded = on ? 1 : 0;
    return TRX_OK;
}

/* diagnostics (include/trx.h): 0 = trx_scenario_evidence evaluates every masked draw to the end */
int trx_set_bounded_evaluation(int mode)
{
    if (mode < 0 || mode > 2) return fail(TRX_ERR_ARG, "bounded evaluation mode must be 0, 1 or 2%s (got %ld)", "", (long)mode);
    g_knob_bounded = mode;
    return TRX_OK;
}

/* tests (include/trx.h): trx_lnl_batch / trx_lnz_scenario evaluate their rows the bounded way, too */
int trx_set_debug_bug(int on)
{
    g_knob_debug_bug = on ? 1 : 0;
    return TRX_OK;
}

/* tests, A/B runs (include/trx.h): rows per wave of the probe pass of a split launch */
int trx_set_probe_rows(int rows)
{
    if (rows < 0 || rows > kCellsMaxRows) return fail(TRX_ERR_ARG, "probe rows must be 0 (automatic) .. 22%s (got %ld)", "", (long)rows);
    g_knob_probe_rows = rows;
    return TRX_OK;
}

int trx_set_debug_bounded_lnl(int on)
{
    g_knob_bounded_lnl = on ? 1 : 0;
    return TRX_OK;
}

/* diagnostics (include/trx.h): 0 = Gauss nodes everywhere, no centre-value stencil */
int trx_set_stencil(int on)
{
    g_knob_stencil = on ? 1 : 0;
    return TRX_OK;
}

/* diagnostics (include/trx.h): 0 = full Kepler solve per sub-exposure */
int trx_set_kepler_stepping(int on)
{
    g_knob_kepler_stepping = on ? 1 : 0;
    return TRX_OK;
}

/* diagnostics (include/trx.h): 0 = evaluate all S sub-exposures of every cell */
int trx_set_supersample_tiers(int on)
{
    g_knob_tiers = on ? 1 : 0;
    return TRX_OK;
}

/* diagnostics (include/trx.h): trx_flux_grid writes the number of model
   evaluations planned for each cell (0, a reduced node count, or nsupersample) instead of the flux */
int trx_set_debug_node_counts(int on)
{
    g_knob_debug_nodes = on ? 1 : 0;
    return TRX_OK;
}

/* (tests) scratch buffers of captured calls: how many a live graph still owns, how many wait in the pool for reuse */
int trx_debug_capture_buffers(long* live, long* idle)
{
    long a = 0, b = 0;
    trx::capture_scratch_stats(&a, &b);
    if (live) *live = a;
    if (idle) *idle = b;
    return TRX_OK;
}

#endif  // TRX_TESTING

#ifdef TRX_CENSUS
int trx_debug_census(unsigned long long* out32, int reset)
{
    TRX_HIP(hipDeviceSynchronize());
    if (out32) TRX_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(trx::g_census), 32 * sizeof(unsigned long long)));
    if (reset) {
        static const unsigned long long zeros[32] = {};
        TRX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(trx::g_census), zeros, sizeof(zeros)));
    }
    return TRX_OK;
}
#endif

#ifdef TRX_PHASE_TIMERS
/* debug builds only: read (and clear) the phase cycle counters */
int trx_debug_phase_cycles(unsigned long long* out8)
{
    TRX_HIP(hipDeviceSynchronize());
    TRX_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_cycles), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    TRX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof(zero)));
    return TRX_OK;
}
#endif

/* frees every per-stream scratch buffer of the library (all devices); the streams must be idle */
int trx_release_scratch(void)
{
    std::lock_guard<std::mutex> lock(trx::g_scratch_mu);
    int cur = 0;
    TRX_HIP(hipGetDevice(&cur));
    for (auto& kv : trx::g_scratch) {
        TRX_HIP(hipSetDevice(kv.first.first));
        for (int sl = 0; sl < trx::kScratchSlots; ++sl)
            if (kv.second.p[sl]) (void)(sl == 3 ? hipHostFree(kv.second.p[sl]) : hipFree(kv.second.p[sl]));
        for (int k = 0; k < trx::kStageRing; ++k) {
            if (kv.second.stage[k]) (void)hipHostFree(kv.second.stage[k]);
            if (kv.second.stage_ev[k]) (void)hipEventDestroy(kv.second.stage_ev[k]);
        }
    }
    trx::g_scratch.clear();
    TRX_HIP(hipSetDevice(cur));
    TRX_HIP(trx::capture_scratch_release_idle());
    return TRX_OK;
}

const char* trx_version(void) { return "triceratops_amd libtrx 0.2.0 (gfx950)"; }
const char* trx_last_error(void) { return g_err; }

int trx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
