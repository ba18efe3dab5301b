// The likelihood path's device code: rowc_kernel, sec_scan_kernel, cells_kernel and the small kernels of the bounded
// evaluation's passes (pilot_stats_kernel, depth_screen_kernel), with the launch arguments and the host / device rules
// they share (RowsArgs, batch_rows, batch_plan, set_scratch).  Included by trx_kernels.hip only, which holds the host
// side: the launch plan, the launchers, the scratch pools and the C ABI.  (Until round 6 all of it was one 3 500-line
// translation unit.)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include <cstddef>
#include <cstdint>

#include "../../include/trx.h"
#include "trx_device.hpp"
#include "trx_internal.hpp"

namespace {

using namespace trx;

constexpr int MODE_LNL = 0;
constexpr int MODE_GRID = 1;

struct RowsArgs {
    int model, flags;
    const double* time;
    const double* flux;
    int n_time;
    double sigma;
    const double* params;
    long n;
    double exptime;
    int S;
    double* out;       // MODE_LNL: [n] chi2/2 ; MODE_GRID: [n][n_time]
    double* out_sec;   // MODE_GRID: [n] or null
    int B;
    long nbatch;
    int use_tiers, debug_nodes;
    int debug_bug;     // trx_set_debug_bug (tests): see cells_entry
    int need_sec;      // rowc_kernel: run the secondary-eclipse scan (EB rows whose depth is used)
    int mark_unwritten;   // rowc_kernel (likelihood launches): fill out[] with the "never written" mark (kUnwrittenBits)
    int use_stencil;   // 0: no stencil.  rowc_kernel looks for a dense uniform time grid (centre-value stencil,
                       // cells_kernel<LONG>); 1: both instantiations of cells_kernel are enqueued and the one
                       // that does not apply returns; 2: only the instantiation the memo predicts is enqueued
    int skip_excl;     // cells_kernel<lnl>: rows the EB secondary rule excludes (+inf whatever the model) are not evaluated
    int* memo;         // rowc_kernel: where to leave the verdict for later launches on this light curve (or null)
    int tl_off;        // cells_kernel: offset (in doubles) of the staged light curve in LDS (shared by the workgroup's waves)
    int wave_off, wave_doubles;   // cells_kernel: offset of the first wave's own LDS block and the size of one (in doubles)
    int probe_rows;    // cells_kernel, probe pass of a split launch: rows per wave its LDS layout holds (0: as the other passes)
    int wave_floor;    // ... the fewest waves the few-rows rule leaves the probe pass (chains: per branch)
    int wave_floor3;   // ... and the survivors' pass
    double* rowc;      // cells_kernel: row constant blocks [n][kRowDoubles] written by rowc_kernel
    int* scan_list;                      // rowc_kernel -> sec_scan_kernel: the rows whose secondary-eclipse verdict is open
    unsigned long long* scan_count;      // ... and their number
    // Rows counted on the device (trx_scenario_evidence: the draws that passed the geometry mask): when
    // n_dev is set the kernels read the row count from it and `n` is only its upper bound (grid, scratch);
    // the rows-per-wave of the batched variant and the batch count follow from the count on the device
    // (same rule as on the host: batch_rows), `B` being the LDS layout's maximum.
    const long* n_dev;
    // Row r of the block is draw src_idx[r] of a [n_param][src_stride] block (the draw kernel's columns by draw index;
    // see `dense` for the layout the scenario path has used since round 5).
    const int* src_idx;
    long src_stride;
    int twin_cols;     // EB_TWIN rows of the draw kernel's block: P = 2 * column 2, a = column 11
    int dense;         // the draw kernel's block holds the masked draws DENSELY (compact_fill_kernel): row r of the block is
                       // column position r (twin rows: src_stride - 1 - r), not draw src_idx[r]; the prior per draw likewise
    int forced_B;      // trx_set_rows_per_wave
    // Bounded evaluation (trx_scenario_evidence only; cells_kernel<..., PRUNE>): see cells_body
    int prune;         // 1: rows that provably carry no weight in the evidence and cannot be its best draw are abandoned
    int pstride;       // every pstride-th time stamp of a row is a probe cell (evaluated first)
    int pstride3;      // the survivors' pass of batches (part 3): its own first phase, every pstride3-th stamp; 0 = one phase
    double prune_c0;   // -ln(2 pi)/2 - ln sigma: log-weight of a row = prune_c0 - chi^2/2 + lnprior
    const double* prune_lp;   // lnprior per DRAW (indexed through src_idx), or null
    int part;          // PRUNE: 1 = the pilot rows [0, min(n, kPilotRows)), 2 = the rows behind them, 0 = all rows;
                       // split (batches of short light curves): 2 = the PROBE pass over the rows behind the pilot (verdict, the
                       // rows still alive listed), 3 = the listed rows evaluated to the end
    int split;
    int* surv_list;                      // split: the rows the probe pass left alive (any order) ...
    unsigned long long* surv_count;      // ... and their number (zeroed by pilot_stats_kernel)
    int* probe_list;                     // split: the rows behind the pilot that the depth screen did not settle
    unsigned long long* probe_count;     // (depth_screen_kernel; zeroed by pilot_stats_kernel)
    TierHead tiers;                      // node counts and radii of the Gauss tiers ...
    const double* tier_xw;               // ... and their (node offset, weight) pairs in device memory (tier_device)
    // wave-uniform fp64 values precomputed on the host: fp64 arithmetic has no scalar unit, so
    // computing them in the kernel parks them in long-lived vector registers
    double s2, rs2, dS, rS;      // sigma^2, 1 / sigma^2, S, 1 / S
};

// One launch chain for several lnZ_* branches (trx_star_enqueue, trx_scenario.hip): the kernels of the likelihood path
// take the branch as the grid's SECOND dimension.  What the branches of a chain share -- the time stamps, the launch
// geometry, the LDS layout, the node tables -- rides in one RowsArgs; what differs sits in a table of small blocks in
// the SAME argument buffer (scalar loads at a dynamic offset: still scalar registers, pointers still known to be global
// memory, no upload), and a kernel patches a local copy of the common block with the entry of its blockIdx.y.
#ifndef TRX_CHAIN_MAX_BRANCHES
#define TRX_CHAIN_MAX_BRANCHES 24
#endif
struct BranchArgs {
    int model, flags, twin_cols, need_sec;
    const double* flux;
    double sigma, s2, rs2, prune_c0;
    const double* params;
    double* out;
    const long* n_dev;
    const int* src_idx;
    const double* prune_lp;
    double* scratch;           // [lists | row blocks | launch header] of this branch (set_scratch)
    unsigned long long* scan_count;   // the scan's persistent counter, at a place that no other layout ever uses
};
constexpr int kChainMaxBranches = TRX_CHAIN_MAX_BRANCHES;
struct BranchTab {
    BranchArgs b[kChainMaxBranches];
};
static_assert(sizeof(RowsArgs) + sizeof(BranchTab) + 16 <= 4096, "kernel argument buffer");

// a wave-uniform double moved to a scalar register pair
__device__ __forceinline__ double uniform(double v)
{
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long uniform_long(long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long)(((unsigned long long)hi << 32) | lo);
}

// bit i (a lane's own index) of a wave-uniform 64-bit mask: the half is chosen first, so the mask stays in its scalar
// register pair (a 64-bit shift by a lane's amount wants it in two VGPRs, held for as long as the mask lives)
__device__ __forceinline__ bool mask_bit(unsigned long long m, int i)
{
    const unsigned w = (i & 32) ? (unsigned)(m >> 32) : (unsigned)m;
    return ((w >> (i & 31)) & 1u) != 0u;
}

// where row `row` of a launch sits in its parameter block / prior array (RowsArgs::params, prune_lp)
__device__ __forceinline__ long row_pos(const RowsArgs& a, long row)
{
    if (a.dense) return a.twin_cols ? a.src_stride - 1 - row : row;
    return a.src_idx ? (long)a.src_idx[row] : row;
}

// number of set bits of `m` below this lane
__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Optional phase timers (build with -DTRX_PHASE_TIMERS; profiles/phase_cycles2.py): wave cycles spent
// loading the row blocks (0), in the window pass (1), the cell plans (2), the pair trips (3) and the
// rest of a chunk (5), summed over all waves.  Not compiled into the product library; the cycle
// counter reads perturb the one-row variant's code heavily (shares only, not times).
#ifdef TRX_PHASE_TIMERS
__device__ unsigned long long g_phase_cycles[8];
#define TRX_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define TRX_TOCK(slot, from) tm[slot] += __builtin_readcyclecounter() - (from)
#else
#define TRX_TICK(var)
#define TRX_TOCK(slot, from)
#endif

// rows whose light curve was not evaluated because lnL_EB_p's secondary-eclipse rule excludes them
// anyway (statistics for benchmarks: trx_skipped_rows)
// rows abandoned by the bounded evaluation (cells_kernel<PRUNE>): statistics, trx_pruned_rows.
// Both counters are 256 shards, 128 bytes apart, and a wave adds its total ONCE, when it leaves: a device-scope atomic
// per batch on one address was ~23 ns each at the memory side -- 17 000 batches: 0.4 ms behind a 0.5 ms kernel (found in
// round 4 under the bounded evaluation's probe pass, which it made slower than the full evaluation).
constexpr int kStatShards = 256, kStatPad = 16;
__device__ unsigned long long g_row_stats[2][kStatShards][kStatPad];       // [0] skipped, [1] abandoned
__device__ __forceinline__ void add_row_stat(int which, unsigned count)
{
    if (count) atomicAdd(&g_row_stats[which][(blockIdx.x + 41u * blockIdx.y) & (kStatShards - 1)][0], (unsigned long long)count);
}

// radius-ratio rule of the reference (likelihoods.py:122-123 scalar, :406/:418 vector)
__device__ __forceinline__ double k_rule(double k, bool scalar_rule)
{
    if (scalar_rule) { if (fabs(k - 1.0) < 1e-6) k *= 0.999; }
    else             { if ((k - 1.0) < 1e-6) k *= 0.999; }
    return k;
}

// ---------------------------------------------------------------------------------------
// The likelihood path: rowc_kernel + cells_kernel.
// Round 1's kernel took one row at a time, 64 consecutive time stamps per trip, every lane walking
// its own cell's nodes.  That loses lanes four ways, the more the shorter the light curve (the
// reference's real operating point is 100-200 binned points, examples/TSCIII_tutorial.ipynb cell
// 4): the per-row prologue runs on 1-4 of the 64 lanes (a third of the wave's cycles at 100 points,
// profiles/r01/r_phase_cycles.txt); a 100-point row fills 64 + 36 lanes; a 64-cell chunk of a
// coarse time grid spans 0.3 d, so in- and out-of-transit cells share every chunk and the lanes of
// the out-of-window cells idle through the plan and the orbit stage; and neighbouring cells have
// different node counts (3-9 Gauss nodes, S next to a contact), so a per-lane node loop runs to
// the largest.  Here
//   * rowc_kernel derives the row constants of 64 rows per workgroup -- lanes = rows, then lanes =
//     (row, point) for the secondary-eclipse scan -- and stores the 19-double blocks in device
//     scratch (152 B per row of the library's per-stream scratch);
//   * cells_kernel takes one row (LONG, 320 points and more) or a batch of B <= 22 rows per wave.
//     The (row, time) cells form ONE index space, cell = r * n_time + j, walked in windows of
//     cells_window() cells: pass 1 applies the transit-window test to 64 cells at a time (across row
//     boundaries), settles the out-of-window cells (model exactly 1) and files the in-window ones,
//     in order, in a list in LDS; pass 2 takes that list 64 cells at a time: each lane plans its
//     cell (Kepler solve at the exposure centre + node count), then the (cell, node) PAIRS of the
//     chunk are dealt to all 64 lanes: a pair steps the orbit from its cell's centre solution,
//     evaluates the Mandel-Agol flux and adds its term to the cell's sum in LDS.  Every stage runs
//     full lanes whatever the mix of node counts.  Cells next to a limb contact (all S
//     sub-exposures, some of them off the disc) are filed again and take a second sweep, so that
//     the first sweep's pairs are all on the disc;
//   * batches: a lane's row constants come from the row blocks in LDS (cells of different rows share
//     a wave); chi^2 of a row = chi^2 of the flat model (every cell exactly 1: one number per
//     launch) + the corrections ((f-m)^2 - (f-1)^2)/sigma^2 of its in-window cells, added to one
//     LDS accumulator per row: the window pass touches no flux, and draws whose model is flat over
//     the data tie EXACTLY (the reference's argsort orders such ties in the best-fit table);
//   * LONG: the row constants ride in scalar registers, time stamps and fluxes are read from global
//     memory, and the lanes sum (f-m)^2/sigma^2 directly (a perfect fit gives exactly 0); a row
//     with a flat model takes the launch's flat-model value, so those rows tie exactly as well.
// Waves per workgroup of the batched variant (rows of short light curves, several per wave).  Every wave works
// through its own batches; what the waves of a workgroup share is read-only LDS -- the staged light curve, the
// node tables -- so four of them hold one copy instead of four and five waves per SIMD fit the CU's 160 KB
// where four did (the variant needs < 102 VGPRs).  One row per wave (LONG): the light curve stays in global
// memory, nothing to share, one wave per workgroup.
#ifndef TRX_BATCH_WAVES
#define TRX_BATCH_WAVES 4
#endif

constexpr int kBatchWaves = TRX_BATCH_WAVES;
__host__ __device__ constexpr int cells_waves(bool long_rows) { return long_rows ? 1 : kBatchWaves; }
// (experiment, profiles/r05: 1 = one row per wave reads its row constants from LDS like the batched variant instead of
// holding them in ~30 scalar registers)
// which instantiations take the arctangent's range constants from the LDS table (the others select them)
// plan_cell reads the tiers' radii and node counts from LDS (true) or from the kernel's arguments
#ifndef TRX_TIER_LDS
#define TRX_TIER_LDS(PRUNE, LONG) true
#endif
// How an instantiation takes the arctangent's range constants (ma_flux<TAB>): 1 = from the LDS table, known at compile
// time (the select-based version is not even in the binary: until round 5 both were, behind a test of the pointer --
// 100 instructions per instantiation that nobody executed, and with them gone the compiler needs 10-40 fewer scalar
// spills and up to 12 fewer VGPRs); 2 = the test of the pointer at run time.  The batched instantiation of the bounded
// evaluation keeps 2: with 1 the register allocator lands on 98 VGPRs, two over the 96 that five waves per SIMD allow
// (two 8-byte spills outside the loops; tests/test_build_resources.py wants none), with 2 on 93.
#ifndef TRX_ATAN_TAB
#define TRX_ATAN_TAB(PRUNE, LONG) (((PRUNE) && !(LONG)) ? 2 : 1)
#endif
#ifndef TRX_LONG_ROWS_IN_LDS
#define TRX_LONG_ROWS_IN_LDS 0
#endif
// 1: a chunk hands the cells whose pairs would leave its last trip mostly empty to the next chunk (cells_body)
#ifndef TRX_CARRY_CELLS
#define TRX_CARRY_CELLS 1
#endif
#ifndef TRX_CELLS_WAVES_PER_EU
#define TRX_CELLS_WAVES_PER_EU 4
#endif
#ifndef TRX_LAZY_TIERS
#define TRX_LAZY_TIERS 1
#endif
#ifndef TRX_CELLS_PAIRS
#define TRX_CELLS_PAIRS 640
#endif
#ifndef TRX_CELLS_WINDOW
#define TRX_CELLS_WINDOW 2048
#endif
constexpr int kCellsMaxRows = 22;
// cells per window pass (in-window list in LDS): 2048 with one row per wave (1024 until round 5: a 2000-point row was two
// window passes, and the cells next to the seam lost their stencil neighbours -- 5.28 -> 5.14 ms per launch of config 1,
// irregular stamps +1.6 %, 1536: half of that; sixteen one-wave workgroups of 9.9 KB still fit a CU); 768 for batches, whose
// ~640 cells fit one window -- the smaller list lets one more wave onto a CU at 200-300 points
// (18.5 -> 17.5 ms per 18 launches at 200 points; 1000-point rows lose 2 % with it)
#ifndef TRX_CELLS_WINDOW_BATCH
#define TRX_CELLS_WINDOW_BATCH 768
#endif
#ifndef TRX_BATCH_WAVES_PER_EU
#define TRX_BATCH_WAVES_PER_EU 5
#endif
constexpr int kCellsWindowLong = TRX_CELLS_WINDOW, kCellsWindowBatch = TRX_CELLS_WINDOW_BATCH;
__host__ __device__ constexpr int cells_window(bool long_rows) { return long_rows ? kCellsWindowLong : kCellsWindowBatch; }
constexpr int kCellsPairs = TRX_CELLS_PAIRS;       // (cell, node) pairs per pass (pair table in LDS)

// Rows per wave of the batched variant: about 640 cells per wave, at most kCellsMaxRows rows.  Measured
// (profiles/r02/g_cells_batch_sweep.txt): 12 rows at 50 points, 6 at 100, 3-4 at 200, 2 at 400 -- larger
// batches fill the chunks better, but the batches of a launch differ in work (rows with long transits),
// and with fewer, longer waves the last round over the chip's wave slots leaves more of them idle.
// Few rows: fewer per wave, down to ~3200 waves a launch (with the tapered plan below the best rows per wave at 100
// points are 3 for 10 000 rows and 6 from 20 000 on, profiles/r04/few_rows_sweep.txt; until then the rule asked for
// 10 000 waves and 30 000 rows ran 6 % slower at 3 per wave than at 4-6).  `n` < 0: the largest value any n gives
// (LDS layout).
__host__ __device__ inline int batch_rows(long n, int n_time, int forced, int floor = 3200)
{
    int B = (640 + n_time / 2) / (n_time > 0 ? n_time : 1);
    B = B < 1 ? 1 : (B > kCellsMaxRows ? kCellsMaxRows : B);
    if (forced > 0) return forced > kCellsMaxRows ? kCellsMaxRows : forced;
    if (n >= 0)
        while (B > 1 && n / B < floor) B = (B + 1) / 2;
    return B;
}

// The batches of a launch, dealt to the XCDs and tapered towards the end of the launch.  Workgroups are dispatched in
// blockIdx order, round the 8 XCDs; an XCD's waves take consecutive positions of its share of the rows (whose blocks
// then stay in that XCD's L2).  The chip holds 5120 waves of the batched variant (5 per SIMD), a launch of 10^5 rows
// at six rows per wave is 3.3 rounds over those slots, and the slots that finish their last batch first idle until
// the last wave of the launch is done: 12 % of the launch (300 000 rows run at 2.35e10 cells/s where 100 000 run
// at 2.04e10: the untapered sweeps of profiles/r04/ab_taper.txt).  So the last positions of every XCD take fewer rows: half a
// slot's worth of positions at half the rows per wave, then as many at a quarter -- the work still out when the slots
// start to drain comes in pieces a quarter the size (100 points, 10^5 rows: 2.04e10 -> 2.14e10 cells/s).
// Positions 0 .. P-1 per XCD; one rule for the host (grid), the workgroup exit test and the batch loop.  Not for the
// passes of the bounded evaluation (`taper` false): a probe pass has so little to do per row that more, smaller
// waves cost more than the tail they fill (+10 % on a 200-point call, profiles/r04/ab_taper.txt).
#ifndef TRX_TAPER_SLOTS
#define TRX_TAPER_SLOTS 320
#endif
struct BatchPlan {
    long R;            // rows per XCD (a multiple of B)
    long ra, rb;       // rows of the XCD's share at B rows per wave, then at B2 (the rest at B3)
    long pa, pb, P;    // positions: [0, pa) B rows, [pa, pa + pb) B2 rows, [pa + pb, P) B3 rows
    int B, B2, B3;
};
__host__ __device__ inline BatchPlan batch_plan(long rows, int B, bool taper = true)
{
    BatchPlan p;
    p.B = B;
    p.B2 = (B + 1) / 2;
    p.B3 = B / 4 > 0 ? B / 4 : 1;
    const long nb = (rows + B - 1) / B;
    p.R = ((nb + 7) / 8) * B;
    long rb = 0, rc = 0;
#ifndef TRX_NO_TAPER
    if (B > 1 && taper) {
        // an XCD holds 640 waves of this kernel (32 CUs x 4 SIMDs x 5); measured per 18 launches of 10^5 rows x 100
        // points: no taper 8.86-8.96 ms, 2560 positions per tier 8.85-8.90, 1280 8.67-8.74, 640 8.51-8.55, 320 8.49-8.52
        // (profiles/r04/ab_taper.txt)
        constexpr long kSlotsPerXcd = TRX_TAPER_SLOTS;
        rc = kSlotsPerXcd * p.B3;
        if (rc > (3 * p.R) / 20) rc = (3 * p.R) / 20;
        rb = kSlotsPerXcd * p.B2;
        if (rb > (3 * p.R) / 10) rb = (3 * p.R) / 10;
        rc -= rc % B;
        rb -= rb % B;
    }
#endif
    p.ra = p.R - rb - rc;
    p.rb = rb;
    p.pa = p.ra / B;
    p.pb = (rb + p.B2 - 1) / p.B2;
    p.P = p.pa + p.pb + (rc + p.B3 - 1) / p.B3;
    return p;
}
// position `pos` of XCD `xcd`: first row (relative to the launch's first) and the rows of the batch there
__host__ __device__ inline void batch_at(const BatchPlan& p, long xcd, long pos, long& off, int& rows)
{
    if (pos < p.pa) { off = pos * p.B; rows = p.B; }
    else if (pos < p.pa + p.pb) {
        off = p.ra + (pos - p.pa) * p.B2;
        const long left = p.ra + p.rb - off;
        rows = (int)(left < p.B2 ? left : p.B2);
    } else {
        off = p.ra + p.rb + (pos - p.pa - p.pb) * p.B3;
        const long left = p.R - off;
        rows = (int)(left < p.B3 ? left : p.B3);
    }
    off += xcd * p.R;
}

// Launch header behind the row blocks in scratch: [0] chi^2 of the flat model, [1] stencil radius
// (0 = no stencil), [2 .. 2 + 2 kStM] stencil weights.
//
// Centre-value stencil (dense uniform time grids; cells_kernel<LONG>).  The reference averages the
// model over S sub-exposures spanning `exptime`.  On a uniform grid whose spacing dt is a fraction
// of the exposure (BASELINE config 1: dt = 0.18 exptime) neighbouring exposures overlap, and where
// the model is analytic the instantaneous flux at the exposure CENTRES of 2 kStM + 1 neighbouring
// cells determines it over the whole exposure: the S-point average of the degree-2 kStM interpolant
// through those centre values is a fixed weighted sum, W_i = (1/S) sum_s l_i(x_s), l_i the Lagrange
// basis on the integer nodes -kStM .. kStM and x_s the sub-exposure offsets in units of dt.  One
// model evaluation per cell instead of 3-4 Gauss nodes.  Error (Cauchy): for a model analytic in a
// disc of radius rho around the cell's centre, |error| <= M max_s prod_i |x_s - i| (dt / (rho - kStM
// dt))^(2 kStM + 1), M the largest flux deficit on the disc (<= 1); the radius below makes that
// 1e-15.  plan_cell tests it exactly like a Gauss tier (no limb contact, real or complex, inside
// the disc).  Anything else -- non-uniform stamps, coarse grids, cells near a contact, the first
// and last kStM cells of a chunk -- takes the Gauss nodes as before.
// Half-width of the stencil.  Round 6: 8 (17 centre values) instead of 6: Cauchy's bound falls with the degree faster
// than the disc must grow with the span -- radius 17.7 -> 13.4 half exposures on config 1's grid (dt = 0.18 exptime), so
// the stencil reaches 15 cells closer to every contact, where the cells took 4-5 Gauss nodes: evaluations per cell 1.184
// -> 1.126, config 1 4.54 -> 4.45 ms.  10 would gain another 0.005 evaluations per cell and lose 6 %: the stencil state of
// a wave is then 64 B more and its 10.2 KB of LDS no longer fit sixteen times into a CU's 160 KB (profiles/r06/ab_stm.txt).
#ifndef TRX_ST_M
#define TRX_ST_M 8
#endif
constexpr int kStM = TRX_ST_M;
constexpr int kHdrFlat = 0, kHdrStRadius = 1, kHdrStW = 2, kHdrHmin = kHdrStW + 2 * kStM + 1, kHdrXmax = kHdrHmin + 1,
              kHdrProbe = kHdrHmin + 2;
// Depth screen of the bounded evaluation (behind the slots above): a model whose flux deficit never exceeds d cannot
// fit the data points that lie deeper than that, whatever its timing: chi^2 >= G(d) = sum_j max(0, (1 - d) - f_j)^2 /
// sigma^2, a function of the light curve alone.  kHdrGrid holds 64 log-spaced depths (1e-5 .. 1), kHdrG the 64 values
// G(depth); a row's largest possible deficit follows from its constants (depth_bound), and G at the next grid depth
// above it is a lower bound of the row's chi^2 before a single cell is looked at.  Most prior draws of a planet
// scenario are too small for a detected signal: this settles them.
constexpr int kHdrGrid = kHdrProbe + 1, kHdrG = kHdrGrid + 64, kHdrDoubles = kHdrG + 64;
__host__ __device__ inline double depth_grid(int i) { return pow(10.0, -5.0 + 5.0 * (double)i / 63.0); }
// Bounded evaluation, several launches (see cells_body, PRUNE, and launch_cells): the first kPilotRows rows are
// evaluated to the end -- they give the launch's running bounds their first values, so that the bound bites from the
// first wave of the later passes on, and they tell whether probing pays at all: pilot_stats_kernel switches it off
// (header slot kHdrProbe) when few pilot rows lie far above the pilot's best.
// (1024 since round 5: with the calls of a target in one launch chain the pilot is a launch of 18 x kPilotRows one-row waves
// that nothing overlaps; 512 / 1024 / 2048 / 4096 rows: 64-TOI step 0.139 / 0.139 / 0.144 / 0.155 s, 75-scenario
// calc_probs 17.6 / 17.6 / 18.0 / 18.6 ms -- profiles/r05/ab_pilot_rows.txt)
#ifndef TRX_PILOT_ROWS
#define TRX_PILOT_ROWS 1024
#endif
constexpr long kPilotRows = TRX_PILOT_ROWS;
// rows per wave of the pilot of a split launch (batches of short light curves)
#ifndef TRX_PILOT_B
#define TRX_PILOT_B 1
#endif
constexpr int kPilotB = TRX_PILOT_B;
// Light curves shorter than this are evaluated in full: with fewer than three stamps per probe cell there is nothing to
// probe.  (Until the fuzz of profiles/fuzz_bounded.py such a launch still ran the passes with a probe stride of 1: the
// probe pass then declined to probe while the third pass waited for its list -- rows behind the pilot were never
// written whenever the pilot's verdict was "probing pays".  tests/test_gpu_bounded.py::test_very_short_light_curves...)
constexpr int kProbeMinPoints = 48;
constexpr int kProbeCells = 16;
constexpr int kThirdStride = 3;          // the survivors' pass of batches takes every third stamp first (0: one phase; see cells_body;
                                         // 64-TOI step 0.1240 -> 0.1211 s, 9371 fuzz configurations clean: profiles/r05/ab_third_stride.txt)
static_assert(kHdrStW + 2 * kStM + 1 <= kHdrDoubles, "launch header");

// Undecided-row list of the secondary-eclipse scan, in FRONT of the row blocks in scratch (a place that does not
// depend on the row count, which may only be known on the device): one 64-bit counter, then one int per row.
__host__ __device__ inline size_t scan_list_doubles(long n_upper) { return 2 + (size_t)(n_upper + 1) / 2; }
// A likelihood launch's scratch: [scan counter, scan list | (bounded evaluation of batches: survivor counter, survivor list,
// probe counter, probe list) | row blocks | launch header], for a.n rows at most (a.split set).
__host__ __device__ inline size_t launch_scratch_doubles(long n_upper, bool split)
{
    return scan_list_doubles(n_upper) * (split ? 3 : 1) + (size_t)n_upper * kRowDoubles + kHdrDoubles;
}
__host__ __device__ inline void set_scratch(RowsArgs& a, double* scratch)
{
    const size_t list_doubles = scan_list_doubles(a.n);
    const bool split = a.split != 0;
    a.scan_count = reinterpret_cast<unsigned long long*>(scratch);
    a.scan_list = reinterpret_cast<int*>(scratch + 2);
    a.surv_count = split ? reinterpret_cast<unsigned long long*>(scratch + list_doubles) : nullptr;
    a.surv_list = split ? reinterpret_cast<int*>(scratch + list_doubles + 2) : nullptr;
    a.probe_count = split ? reinterpret_cast<unsigned long long*>(scratch + 2 * list_doubles) : nullptr;
    a.probe_list = split ? reinterpret_cast<int*>(scratch + 2 * list_doubles + 2) : nullptr;
    a.rowc = scratch + list_doubles * (split ? 3 : 1);
}
// the arguments of branch blockIdx.y of a chain (see BranchArgs): the common block patched with the branch's own
__device__ __forceinline__ RowsArgs star_args(const RowsArgs& common, const BranchTab& bt, int part, int branch = -1)
{
    RowsArgs a = common;
    const BranchArgs& b = bt.b[branch < 0 ? (int)blockIdx.y : branch];
    a.model = b.model; a.flags = b.flags; a.twin_cols = b.twin_cols; a.need_sec = b.need_sec;
    a.flux = b.flux; a.sigma = b.sigma; a.s2 = b.s2; a.rs2 = b.rs2; a.prune_c0 = b.prune_c0;
    a.params = b.params; a.out = b.out; a.n_dev = b.n_dev; a.src_idx = b.src_idx; a.prune_lp = b.prune_lp;
    a.part = part;
    set_scratch(a, b.scratch);
    a.scan_count = b.scan_count;
    return a;
}

// The constants of one row (lanes = rows): unit conversion (likelihoods.py:337-347, 399-411), the radius-ratio rule,
// orbit constants and transit window, dilution, limb weights -> c.  EB rows whose secondary depth is used (need_sec)
// also get the orbit of the secondary eclipse (sc) and the two dilution constants of its depth.
// PRIMARY = false: only the secondary orbit and the dilution constants (sec_scan_kernel; c is left untouched).
template <bool PRIMARY = true>
__device__ __forceinline__ void row_constants(const RowsArgs& a, const long n, const long row, RowC& c, const bool want_sec,
                                              RowC& sc, double& ysec, double& fdil)
{
    const bool eblike = (a.model == TRX_MODEL_EB) || (a.model == TRX_MODEL_EB_TWIN);
    const bool is_host = (a.flags & TRX_FLAG_COMPANION_IS_HOST) != 0;
    const bool scalar_k = (a.flags & TRX_FLAG_SCALAR_K) != 0;
    // row r = column r of the [n_param][n] block, or draw src_idx[r] of a [n_param][src_stride] one
    const double* p = a.params + row_pos(a, row);
    const long ps = (a.dense || a.src_idx) ? a.src_stride : n;
    double u1, u2;
    ysec = 0.0;
    fdil = 0.0;
    if (a.model == TRX_MODEL_RAW) {
        u1 = p[7 * ps]; u2 = p[8 * ps];
        if (PRIMARY) {
            orbit_init(c, p[0], p[1 * ps], p[2 * ps], p[3 * ps], p[4 * ps], p[5 * ps], p[6 * ps], a.exptime);
            c.rdil = 1.0;
        }
    } else {
        double k, ksec = 0.0, per, inc, acm, R_s, e, argp, comp_fr, feb = 0.0;
        if (a.model == TRX_MODEL_TP) {
            const double R_p = p[0];
            per = p[1 * ps]; inc = p[2 * ps]; acm = p[3 * ps]; R_s = p[4 * ps];
            u1 = p[5 * ps]; u2 = p[6 * ps]; e = p[7 * ps]; argp = p[8 * ps]; comp_fr = p[9 * ps];
            k = R_p * kRearth / (R_s * kRsun);                      // likelihoods.py:340
        } else {
            const double R_EB = p[0], eb_fr = p[1 * ps];
            per = p[2 * ps]; inc = p[3 * ps]; acm = p[4 * ps]; R_s = p[5 * ps];
            u1 = p[6 * ps]; u2 = p[7 * ps]; e = p[8 * ps]; argp = p[9 * ps]; comp_fr = p[10 * ps];
            if (a.twin_cols) {                   // marginal_likelihoods.py:300-339: twice the period and its a
                per = per * 2.0;
                acm = p[11 * ps];
            }
            feb = eb_fr / (1.0 - eb_fr);                            // :401
            k = k_rule(R_EB / R_s, scalar_k);                       // :405-406
            ksec = scalar_k ? (1.0 / k) : k_rule(R_s / R_EB, false); // :137 / :417-418
        }
        const double fcomp = comp_fr / (1.0 - comp_fr);             // :337, :399
        const double a_R = acm / (R_s * kRsun);                     // :343, :409
        const double inc_r = inc * (kPi / 180.0);                   // :344, :410
        const double w = (90.0 - argp) * (kPi / 180.0);             // :345, :411
        if (PRIMARY) orbit_init(c, k, 0.0, per, a_R, inc_r, e, w, a.exptime);
        double xeb = 0.0;
        if (!eblike) {
            fdil = is_host ? (1.0 / fcomp) : fcomp;                 // :352-357
        } else {
            if (want_sec) {
                const double wsec = (90.0 - argp + 180.0) * (kPi / 180.0);  // :419
                orbit_init<false>(sc, ksec, 0.0, per, a_R, inc_r, e, wsec, 0.0);
                const Limb L = limb_weights(u1, u2);
                sc.cle = L.cle; sc.cld = L.cld; sc.ced = L.ced;
                sc.rdil = 1.0; sc.excl = 0.0;
            }
            if (is_host) {                                          // :427-432
                xeb = feb / fcomp;
                ysec = fcomp / feb;
                fdil = 1.0 / (fcomp + feb);
            } else {                                                // :433-438
                xeb = feb / 1.0;
                ysec = 1.0 / feb;
                fdil = fcomp / (1.0 + feb);
            }
        }
        // the two dilution stages as one factor on the flux DEFICIT: (m + x)/(1 + x) = 1 - (1 - m)/(1 + x), so
        // an unocculted point stays exactly 1 and a cell costs one fma instead of two divisions; a flux ratio
        // that is not finite makes the reference's quotient NaN (inf / inf), hence NaN here
        if (PRIMARY) {
            c.rdil = 1.0 / ((1.0 + xeb) * (1.0 + fdil));
            if (!(fabs(xeb) < INFINITY) || !(fabs(fdil) < INFINITY)) c.rdil = NAN;
        }
    }
    if (PRIMARY) {
        const Limb L = limb_weights(u1, u2);
        c.cle = L.cle; c.cld = L.cld; c.ced = L.ced;
        c.excl = 0.0;
    }
}

// The launch's header (flat-model chi^2, stencil verdict, bounds of the bounded evaluation, the scan's counter):
// a workgroup of its own (the last one of rowc_kernel's grid), since a wave doing it before its rows would be the
// launch's long pole on a 2000-point light curve.
__device__ __forceinline__ void launch_header(const RowsArgs& a, const long n)
{
    const int lane = threadIdx.x;
    double* hdr = a.rowc + n * kRowDoubles;
    if (a.flux) {
        // chi^2 of the flat model (every cell exactly 1), one number per launch, behind the row
        // blocks: rows whose model is flat over the data get exactly this value and tie
        double acc = 0.0;
        for (int j = lane; j < a.n_time; j += 64) {
            const double d = a.flux[j] - 1.0;
            acc = fma(d * d, a.rs2, acc);            // (every chi^2 term of the path is (f - m)^2 x (1 / sigma^2): the
                                                     // same operation everywhere, so that flat rows tie exactly)
        }
        acc = wave_sum(acc);
        if (a.prune) {
            // the depth screen's table (bounded evaluation only): lane i takes depth i of the grid; the light curve goes
            // through LDS 512 points at a time (a lane reading flux[j] from memory in a serial loop waited ~200 cycles
            // per point: 190 us on a 2000-point curve, on the critical path of every launch)
            __shared__ double gbuf[512];
            const double dpt = (lane == 63) ? 1.0 : depth_grid(lane);
            const double lim = 1.0 - dpt;
            double g = 0.0;
            for (int j0 = 0; j0 < a.n_time; j0 += 512) {
                const int m = (a.n_time - j0 < 512) ? (a.n_time - j0) : 512;
                __syncthreads();
                for (int j = lane; j < m; j += 64) gbuf[j] = a.flux[j0 + j];
                __syncthreads();
#pragma unroll 8
                for (int j = 0; j < m; ++j) {
                    const double d = lim - gbuf[j];
                    g = fma(d > 0.0 ? d * d : 0.0, a.rs2, g);
                }
            }
            hdr[kHdrGrid + lane] = dpt;
            hdr[kHdrG + lane] = g;
        }
        if (lane == 0) {
            hdr[kHdrFlat] = acc;
            // running bounds of the launch (cells_kernel<PRUNE>): the smallest chi^2/2 and the largest
            // log-weight among the rows finished so far
            hdr[kHdrHmin] = INFINITY;
            hdr[kHdrXmax] = -INFINITY;
            hdr[kHdrProbe] = 1.0;
        }
    }
    {
        // is the time grid uniform and dense enough for the centre-value stencil?
        const int l = lane, nt = a.n_time;
        bool ok = a.use_stencil && nt >= 64 && a.S >= 2 && a.exptime > 0.0;
        double dt = 0.0, t0 = 0.0;
        if (ok) {
            t0 = a.time[0];
            dt = (a.time[nt - 1] - t0) / (double)(nt - 1);
            double dev = 0.0, big = 0.0;
            for (int j = l; j < nt; j += 64) {
                const double t = a.time[j];
                dev = fmax(dev, fabs(t - fma((double)j, dt, t0)));
                big = fmax(big, fabs(t));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                dev = fmax(dev, __shfl_xor(dev, o, 64));
                big = fmax(big, __shfl_xor(big, o, 64));
            }
            // stamps within 4 ulp of t0 + j dt (np.linspace is); NaN stamps fail the comparison
            ok = dt > 0.0 && dev <= 4.0 * 2.220446049250313e-16 * fmax(big, dt);
        }
        // spacing in units of the exposure: at most 0.3 (the exposures must overlap well), and the
        // sub-exposures must stay in the middle of the node span, |x_s| <= 0.5 / u <= kStM / 2 + 0.5,
        // where equispaced interpolation is well conditioned (beyond it the weights grow and alternate)
        const double u = ok ? dt / a.exptime : 1.0;
        ok = ok && u <= 0.3 && 0.5 / u <= 0.5 * kStM + 0.5;
        double w = 0.0, pmax = 0.0;
        if (ok && l <= 2 * kStM + 1) {
            // lanes 0 .. 2 kStM: the weight of node i = l - kStM; lane 2 kStM + 1: max_s prod |x_s - i|
            const int i = l - kStM;
            for (int sidx = 1; sidx <= a.S; ++sidx) {
                const double x = (((double)sidx - 0.5) * a.rS - 0.5) / u;
                double num = 1.0, den = 1.0;
                for (int k = -kStM; k <= kStM; ++k) {
                    if (l == 2 * kStM + 1) num *= fabs(x - (double)k);
                    else if (k != i) { num *= x - (double)k; den *= (double)(i - k); }
                }
                if (l == 2 * kStM + 1) pmax = fmax(pmax, num);
                else w += num / den;
            }
            w *= a.rS;
        }
        pmax = __shfl(pmax, 2 * kStM + 1, 64);
        // conditioning of the rule: sum |W_i| (1 for a positive rule); refuse anything above 3
        double wabs = (l <= 2 * kStM) ? fabs(w) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wabs += __shfl_xor(wabs, o, 64);
        ok = ok && wabs <= 3.0;
        if (l <= 2 * kStM) hdr[kHdrStW + l] = ok ? w : 0.0;
        if (l == 0) {
            // rho / (exptime / 2) = 2 u (kStM + (pmax / 1e-15)^(1 / (2 kStM + 1))), 10 % on top
            double radius = 0.0;
            if (ok) {
                radius = 1.1 * 2.0 * u * ((double)kStM + pow(pmax * 1e15, 1.0 / (2.0 * kStM + 1.0)));
                if (!(radius > 0.0 && radius <= 40.0)) radius = 0.0;
            }
            hdr[kHdrStRadius] = radius;
            if (a.memo) *a.memo = (radius > 0.0) ? 2 : 1;
        }
    }
}

// Row constants of 64 rows per workgroup (one wave: lanes = rows), written to a.rowc[n][kRowDoubles] as one
// contiguous run through LDS.
//
// The secondary eclipse of an EB row (likelihoods.py:417-424: the model over np.linspace(-0.05, 0.05, 25) around the
// secondary conjunction, its minimum diluted into `secdepth`).  Likelihood calls only need the verdict "depth >= 1.5
// sigma" (:535-538), and the scan's minimum can only be deeper than any one of its points: the point at the
// secondary conjunction (the 13th of the 25) settles 97 % of the draws of a typical lnZ_*EB call right here, lanes =
// rows.  The rows it leaves open go on a list in scratch -- one atomic per wave -- and sec_scan_kernel, enqueued
// behind this kernel, scans only those, compacted ACROSS workgroups (until round 3 the scan ran inside this kernel on
// the few open rows of each 64-row block: 256 threads per block of which one wave derived the constants, a serial
// chain of ~2000 fp64 instructions at a quarter of the wave slots -- 232 us per call of a 75-scenario calc_probs).
// When the depth itself is asked for (trx_flux_grid's out_secdepth) every row is scanned: no list, no quick test, and
// the secondary orbit is derived by the scan kernel.
// The mark of a row no likelihood pass has written yet (a quiet NaN with a payload no arithmetic produces): rowc_kernel
// fills a likelihood launch's chi^2 array with it, every pass overwrites its rows, and the reduction of a scenario
// (lme_partial_kernel<SCEN>) reports a row that still carries it -- a row "never written" would otherwise read as
// whatever the stream's previous call left there (two such bugs shipped in round 4).
constexpr unsigned long long kUnwrittenBits = 0x7ff8dead0badc0deull;

template <bool SEC>
__device__ __forceinline__ void rowc_body(const RowsArgs& a)
{
    __shared__ RowC rows_out[64];
    const int lane = threadIdx.x;
    const long n = a.n_dev ? *a.n_dev : a.n;
    const long nblk = (n + 63) / 64;
    // the last workgroup writes the launch header, the others stride over the blocks of rows (with the row count
    // on the device the grid is a guess)
    if (blockIdx.x == gridDim.x - 1) {
        launch_header(a, n);
        return;
    }
    const bool quick = SEC && a.out_sec == nullptr;
    int* list = a.scan_list;
    unsigned long long* count = a.scan_count;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x - 1) {
        const long base = blk * 64;
        const int nb = (int)((n - base < 64) ? (n - base) : 64);
        bool open = false;
        if (lane < nb) {
            RowC c, sc;
            double ysec, fdil;
            row_constants(a, n, base + lane, c, quick, sc, ysec, fdil);
            if (quick) {
#ifndef TRX_SEC_FULL_SCAN
                const Limb L{sc.cle, sc.cld, sc.ced};
                const double ts = __dadd_rn(__dmul_rn(0.1 / 24.0, 12.0), -0.05);       // linspace(-0.05, 0.05, 25)[12]
                const double f = exposure_flux(sc, L, ts, 0.0, 1, 1.0, 1.0, false, a.tiers);
                const double m = (f + ysec) / (1.0 + ysec);
                const double depth = 1.0 - (m + fdil) / (1.0 + fdil);
                // deep enough already (or NaN: np.min would propagate it) -> RowC::excl, :535; else the scan decides
                if (!(depth < 1.5 * a.sigma)) c.excl = 1.0;
                else open = true;
#else
                open = true;
#endif
            }
            // through LDS to memory: 64 x 19 doubles leave the block as one contiguous 9.5 KB run (a lane
            // writing its own 152-byte block made every store instruction touch 64 cache lines)
            const double* src = reinterpret_cast<const double*>(&c);
            double* stage = reinterpret_cast<double*>(&rows_out[lane]);
#pragma unroll
            for (int q = 0; q < kRowDoubles; ++q) stage[q] = src[q];
        }
        if (quick) {
            const unsigned long long mo = __ballot(open);
            if (mo) {
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd(count, (unsigned long long)__popcll(mo));
                at = __shfl(at, 0, 64);
                // (bounded: a counter left non-zero by a call that failed between this kernel and the cells_kernel that
                // resets it must not push the list past its n_upper + 1 entries)
                const unsigned long long slot = at + (unsigned long long)lanes_below(mo);
                if (open && slot <= (unsigned long long)a.n) list[slot] = (int)(base + lane);
            }
        }
        if (a.mark_unwritten && lane < nb) a.out[base + lane] = __longlong_as_double((long long)kUnwrittenBits);
        __syncthreads();
        {
            const double* src = reinterpret_cast<const double*>(rows_out);
            double* out = a.rowc + base * kRowDoubles;
            for (int i = lane; i < nb * kRowDoubles; i += 64) out[i] = src[i];
        }
        __syncthreads();
    }
}

template <bool SEC>
__global__ __launch_bounds__(64) void rowc_kernel(RowsArgs a)
{
    rowc_body<SEC>(a);
}

// chain: branch = blockIdx.y; the branches that need the secondary-eclipse verdict take the SEC body
__global__ __launch_bounds__(64) void rowc_kernel_star(RowsArgs common, BranchTab bt)
{
    const RowsArgs a = star_args(common, bt, 0);
    if (a.need_sec) rowc_body<true>(a);
    else rowc_body<false>(a);
}

// The 25-point scan of the rows rowc_kernel<true> left open (or of every row when the depth itself is asked for):
// E list entries per workgroup of one wave.  Lanes = rows derive the secondary orbit, then the E x 25 (row, point)
// cells are dealt to the lanes -- the points are reached by Newton steps from the secondary conjunction -- the
// minimum is taken with LDS atomics (min ignores NaN, so NaN is flagged separately: np.min propagates it), and
// lanes = rows turn it into the exclusion flag of the row's block / the secdepth output.
// E = 64 when every row is scanned (full lanes throughout).  The open rows of a likelihood call are few -- ~3000 of
// the 1e5 masked draws of a lnZ_*EB call -- and a wave's work is one serial chain (orbit constants, then E x 25 / 64
// model evaluations of ~1000 fp64 instructions each): E = 8 spreads them over eight times the waves (117 -> 30 us).
template <int E>
__device__ __forceinline__ void sec_scan_body(const RowsArgs& a)
{
    __shared__ RowC srows[E];
    __shared__ double secmin[E];
    __shared__ int secnan[E];
    const int lane = threadIdx.x;
    const long n = a.n_dev ? *a.n_dev : a.n;
    const bool all_rows = a.out_sec != nullptr;
    const int* list = a.scan_list;
    const long nu = all_rows ? n : (long)*a.scan_count;
    for (long e0 = (long)blockIdx.x * E; e0 < nu; e0 += (long)gridDim.x * E) {
        const int ne = (int)((nu - e0 < E) ? (nu - e0) : E);
        long row = 0;
        double ysec = 0.0, fdil = 0.0;
        if (lane < ne) {
            row = all_rows ? e0 + lane : (long)list[e0 + lane];
            row = (row >= 0 && row < n) ? row : 0;              // (a list left over by an aborted call cannot reach outside)
            RowC unused;
            row_constants<false>(a, n, row, unused, true, srows[lane], ysec, fdil);
            secmin[lane] = INFINITY;
            secnan[lane] = 0;
        }
        __syncthreads();
        for (int it = lane; it < ne * kSecPoints; it += 64) {
            const int ri = it / kSecPoints, j = it - ri * kSecPoints;
            const RowC sc = srows[ri];
            const Limb L{sc.cle, sc.cld, sc.ced};
            // np.linspace(-0.05, 0.05, 25): start + j*step, last point exact
            double ts = __dadd_rn(__dmul_rn(0.1 / 24.0, (double)j), -0.05);
            if (j == kSecPoints - 1) ts = 0.05;
            const double f = exposure_flux(sc, L, ts, 0.0, 1, 1.0, 1.0, false, a.tiers);
            if (f != f) atomicOr(&secnan[ri], 1);
            else __hip_atomic_fetch_min(&secmin[ri], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (lane < ne) {
            double m = secnan[lane] ? NAN : secmin[lane];                   // np.min propagates NaN
            m = (m + ysec) / (1.0 + ysec);
            const double secdepth = 1.0 - (m + fdil) / (1.0 + fdil);
            a.rowc[row * kRowDoubles + (kRowDoubles - 1)] = (secdepth < 1.5 * a.sigma) ? 0.0 : 1.0;  // RowC::excl, :535
            if (a.out_sec) a.out_sec[row] = secdepth;
        }
        __syncthreads();
    }
}

template <int E>
__global__ __launch_bounds__(64) void sec_scan_kernel(RowsArgs a)
{
    sec_scan_body<E>(a);
}

// (the grid's second dimension runs over the branches that HAVE a secondary-eclipse rule -- six of a target's eighteen:
// launched over all of them, two thirds of the kernel's 70 000 workgroups had nothing to do but be dispatched)
struct BranchMap {
    unsigned char id[TRX_CHAIN_MAX_BRANCHES];
};
__global__ __launch_bounds__(64) void sec_scan_kernel_star(RowsArgs common, BranchTab bt, BranchMap map)
{
    const RowsArgs a = star_args(common, bt, 0, (int)map.id[blockIdx.y]);
    if (!a.need_sec) return;
    sec_scan_body<8>(a);
}
static_assert(offsetof(RowC, excl) == (kRowDoubles - 1) * sizeof(double), "excl is the last field of RowC");

// per-cell state of the chunk in flight (lane = cell), read by the lanes its pairs are dealt to
struct CellState {
    double sE[64], cE[64];                  // eccentric anomaly at the exposure centre
    double t[64];                           // exposure centre
    double facc[64];                        // the cell's sum over its nodes
    unsigned meta[64];                      // row | (tier + 1) << 8 | anchored << 16 (tier -1 = all S sub-exposures): one read per pair
                                            // (| valid << 17 | node count << 18: what a carried cell's next chunk needs, cells_body)
    unsigned short rel[64];                 // the cell's entry in the window's list
};
// centre-value stencil (LONG only): the chunk's centre fluxes and the launch's weights
struct StencilState {
    double fc[64 + ((kStM + 7) & ~7)];                      // [kStM + h]: the centre flux of lane h; [0, kStM): the previous chunk's last owned cells'
    double stw[(2 * kStM + 1 + 7) & ~7];
};
constexpr int kCentreNode = 1023;           // pair table: "the exposure centre itself" in the node field

// What a verdict of the bounded evaluation has to allow for when the cells done so far carry the fp32 flux model
// (ma_flux_f32; the probe pass of a split launch takes it whatever the call's precision -- TRX_PROBE_FP32, trx_kernels.hip --
// because its rows are either abandoned there or evaluated again from the start by the survivors' pass).  With r_j the
// residuals against the fp32 model and |m_fp32 - m| <= eps in each of the n probed in-window cells,
//     sum (r_j + d_j)^2 >= sum r_j^2 - 2 eps sum |r_j| >= sum r_j^2 - 2 eps sqrt(n sum r_j^2),
// and sum r_j^2 / (2 sigma^2) is at most the bound lb itself: chi^2/2 >= lb - eps sqrt(2 n lb) / sigma.  eps = 1e-5, five
// times the 2e-6 tests/test_gpu_batch.py holds the fp32 model to (measured: 5e-7).  At sigma = 5e-4 and lb = 1e4 that is
// 11; a row needs lb > best + 90 to be abandoned.
__device__ __forceinline__ double fp32_model_slack(double lb, int n_probed, double rs2)
{
    return 1e-5 * sqrt(2.0 * (double)n_probed * rs2 * fmax(lb, 0.0));
}

// exclusive prefix sum over the lanes of a non-negative count < 2^BITS, and the wave total
template <int BITS>
__device__ __forceinline__ int lane_prefix(int cnt, int& total)
{
    int off = 0;
    total = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1);
        off += lanes_below(m) << b;
        total += __popcll(m) << b;
    }
    return off;
}

// LONG = false: a batch of B rows per wave, light curve staged in LDS, row constants read from the
//               row blocks in LDS, chi^2 corrections in one LDS accumulator per row;
// LONG = true:  one row per wave (light curves of kCellsLongFrom points and more): row constants in
//               scalar registers, time stamps and fluxes read from global memory (a chunk's cells
//               are mostly neighbours), chi^2 summed directly per lane and reduced once per row.
// ST: the launch uses the centre-value stencil (decided on the device by rowc_kernel: the kernel
// below picks the instantiation, so a launch without it runs exactly the code it ran before)
// The largest flux deficit the model of a row can show at any time (depth screen, kHdrG): a body of radius ratio k
// hides at most the fraction k^2 of the disc, where the intensity is at most Imax against the disc's mean (1 - u1/3 -
// u2/6 for a unit centre); dilution shrinks it by rdil; the exposure average of deficits below a bound stays below it.
__device__ __forceinline__ double depth_bound(const RowC& c)
{
    const double om4 = 1.0 / (c.cle + c.cld);               // limb_weights: cle + cld = 1 / (1 - u1/3 - u2/6)
    const double u2 = c.ced * om4, u1 = c.cld * om4 - 2.0 * u2;
    double imax = fmax(1.0, 1.0 - u1 - u2);                 // 1 - u1 x - u2 x^2 on [0, 1]: the ends ...
    if (u2 > 0.0 && u1 < 0.0) imax += u1 * u1 / (4.0 * u2); // ... and no more than the vertex adds
    // (no cap at 1: a law whose intensity turns negative at the limb -- u1 + u2 > 1 -- lets a large body hide MORE than
    // the whole flux; depth_screen has nothing to say from 1 on)
    const double d = c.k * c.k * imax * (c.cle + c.cld) * c.rdil * (1.0 + 1e-12);
    return (d >= 0.0) ? d : INFINITY;                       // (NaN anywhere: no bound)
}

// chi^2 / 2 that a row whose deficit never exceeds `d` cannot go below (0 when the table has nothing to say)
__device__ __forceinline__ double depth_screen(const double* hdr, double d)
{
    if (!(d < 1.0)) return 0.0;
    // first grid depth >= d (the grid is log-spaced: an fp32 logarithm finds the neighbourhood, two steps settle it)
    int i = (int)ceilf((log10f((float)fmax(d, 1e-30)) + 5.0f) * (63.0f / 5.0f));
    i = i < 0 ? 0 : (i > 63 ? 63 : i);
    if (i > 0 && hdr[kHdrGrid + i - 1] >= d) --i;
    if (hdr[kHdrGrid + i] < d) i = (i < 63) ? i + 1 : 63;
    if (hdr[kHdrGrid + i] < d) return 0.0;
    const double g = 0.5 * hdr[kHdrG + i];
    return g - fma(1e-9, g, 1e-9);
}

// running bounds of a launch (see PRUNE below): smallest finished chi^2/2, largest finished log-weight
__device__ __forceinline__ void tighten_bounds(double* hdr, double h, double x)
{
    if (h < __hip_atomic_load(&hdr[kHdrHmin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_min(&hdr[kHdrHmin], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x > __hip_atomic_load(&hdr[kHdrXmax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_max(&hdr[kHdrXmax], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The lanes of a wave exchange data through LDS (pair tables, cell state, row accumulators).  LDS operations of
// one wave are issued and performed in order, so all the hand-over needs is that the compiler keeps them in
// program order and re-reads memory afterwards: a fence at wavefront scope, no instruction.  (A workgroup
// barrier would also do for one wave per workgroup, but the waves of the batched variant run independent
// loops of different lengths.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rows per wave of the passes of the bounded evaluation over LISTED rows (batched variant: the probe pass, part 2, and
// the survivors' pass, part 3).  ONE rule for cells_entry -- which workgroups have batches at all -- and cells_body --
// which rows they are: two copies of it that drift apart leave batches that no workgroup evaluates (round 4's "rows
// never written"; round 5 nearly repeated it when the probe pass got its own rows per wave, and the guard of
// lme_partial_kernel said so at once).  `rows`: the pass's row count, from the device.
__device__ __forceinline__ int listed_pass_rows(const RowsArgs& a, long rows, int part)
{
    // few rows: fewer per wave, by the host's rule for a launch of that many rows
    int B = batch_rows(rows, a.n_time, a.forced_B, part == 3 ? a.wave_floor3 : 3200);
    B = B < a.B ? B : a.B;
    // the probe pass: as many as its own LDS layout holds -- only the probe cells are filed --, halved by the same rule
    if (part == 2 && a.probe_rows > 1) {
        B = a.probe_rows;
        while (B > 1 && rows / B < a.wave_floor) B = (B + 1) / 2;
    }
    return B;
}

// PRUNE (trx_scenario_evidence; MODE_LNL, no stencil): bounded evaluation.  The evidence is a sum of
// exp(c0 - chi^2/2 + lnprior) over the rows and the reduction drops every term more than 80 below the
// largest (lme_partial_kernel: it cannot change an fp64 sum that is >= 1); the best draw is the row with
// the smallest chi^2.  chi^2 only grows as cells are added, so a row whose chi^2/2 over the cells done so
// far already (i) exceeds the smallest FINISHED chi^2/2 of the launch and (ii) puts its log-weight 90
// below the largest finished log-weight can neither be the best draw nor carry weight: it is abandoned
// and reports the bound it reached (any value >= it gives the same lnZ bits and the same best draw).
// The two running bounds live behind the launch header (global atomics, min / max: they only tighten, a
// stale read is merely looser).  To let the bound bite early a row's cells are taken in two phases:
// every pstride-th time stamp (the probe cells, ~16 per row) plus -- for free -- all its out-of-window
// cells, then the verdict, then the rest.  On a real detection most prior draws miss the observed depth
// or duration by far: 95 % of the rows of TOI-465.01's lnZ_TTP stop at the probe
// (profiles/prune_potential.py).  Which rows stop depends on timing, the results do not.
// The argument block read again from the kernel-argument segment at the top of every trip of the row loop: scalar loads
// where a field is used, instead of ~60 scalar registers held from the kernel's entry on -- which the compiler parks in
// VGPR lanes (v_writelane / v_readlane: VALU issue, 138 + 103 of them in the one-row stencil instantiation before this,
// 107 + 90 after).  ARGS: 0 = the block as passed, 1 = cells_kernel's argument, 2 = cells_kernel_star's (the common block
// patched with the branch's entries, star_args, derived again: a few dozen scalar loads per trip).
#ifndef TRX_ARGS_RELOAD
#define TRX_ARGS_RELOAD 1
#endif
struct StarKernArgs {              // cells_kernel_star's argument segment
    RowsArgs common;
    BranchTab bt;
    int part;
};
template <typename T>
__device__ __forceinline__ const T* kernel_arguments()
{
    typedef const __attribute__((address_space(4))) T* KernArgs;
    KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));            // (opaque: the loads stay where the fields are used)
    return (const T*)p;
}

template <int MODE, bool STEP, bool FP32, bool LONG, bool ST, bool PRUNE, int ARGS = 0>
__device__ __forceinline__ void cells_body(const RowsArgs& a_in, const double st_radius)
{
    constexpr int kReload = TRX_ARGS_RELOAD ? ARGS : 0;
    const RowsArgs& a = a_in;
    static_assert(!PRUNE || (MODE == MODE_LNL && !ST), "bounded evaluation: likelihood mode, no stencil");
    extern __shared__ double lds_all[];
    // (not in the diagnostic instantiations that solve Kepler's equation per pair; not with one row per wave: 2000 irregular
    // stamps -0.5 %, and the stencil instantiation, which never carries, -2.3 % for the registers the code costs)
    constexpr bool kCarry = TRX_CARRY_CELLS && STEP && !LONG;
    // (the stencil instantiation: tiers on demand, centres evaluated where they were planned -- plan_cell<LAZY>)
    constexpr bool kLazyTiers = TRX_LAZY_TIERS && ST && STEP && LONG && !PRUNE;
    constexpr int W = cells_waves(LONG);
    const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;     // (scalar: so is all that follows from it)
    // shared by the workgroup's waves: node tables, atan constants, (short curves) the light curve
    double* tier_xw = lds_all;
    double* atab = tier_xw + 2 * kTiers * kTierMaxNodes;             // atan_pos_tab's range constants
    double* thead = atab + kAtanRanges * kAtanCols;                  // the tiers' radii and node counts (plan_cell)
    // this wave's own: row blocks, accumulators, pair table, in-window list, cell state
    double* lds = lds_all + a.wave_off + (size_t)wave * a.wave_doubles;
    const int Bl = LONG ? 1 : a.B;                                    // rows the LDS layout holds
    RowC* rows = reinterpret_cast<RowC*>(lds);
    double* hacc = lds + (size_t)Bl * kRowDoubles;                    // [Bl] chi^2 corrections per row
    double* hmout = hacc + Bl;                                        // [Bl] diluted model of an unocculted cell: 1, or NaN
    double* hrem = hmout + Bl;                                        // [Bl] PRUNE: (f - 1)^2 / s2 over the row's in-window cells not done yet
    double* hlp = hrem + Bl;                                          // [Bl] PRUNE: lnprior of the row's draw (read twice per batch: not worth two VGPRs)
    unsigned short* pdesc = reinterpret_cast<unsigned short*>(hlp + Bl);    // [kCellsPairs] pair -> cell lane | node << 6
    constexpr int kCellsWindow = cells_window(LONG);
    unsigned short* winlist = pdesc + kCellsPairs;                    // [kCellsWindow] in-window cells
    CellState& cs = *reinterpret_cast<CellState*>(winlist + kCellsWindow);
    StencilState& ss = *reinterpret_cast<StencilState*>(&cs + 1);        // LONG only (behind the cell state)
    // short curves: the light curve itself in LDS -- every chunk reads time stamps and fluxes of
    // arbitrary cells, and a global load right before its use costs more than the chunk's other
    // "rest" work
    const double* tl = LONG ? a.time : (lds_all + a.tl_off);          // [n_time]
    const double* fl = LONG ? a.flux : (tl + a.n_time);               // [n_time] (MODE_LNL)
    // the node tables into LDS as (node offset, weight) pairs (one 16-byte read per pair), an entry per lane: a
    // loop on one lane was 420 wave instructions per workgroup -- 3 % of a batch at 100 points
    if (a.use_tiers) {
        for (int i = threadIdx.x; i < 2 * kTiers * kTierMaxNodes; i += 64 * W) tier_xw[i] = a.tier_xw[i];
    }
    if (threadIdx.x < kAtanRanges * kAtanCols) atab[threadIdx.x] = kAtanTable[threadIdx.x];
    if (threadIdx.x == 0) {
        // (literal indices: see plan_cell)
        thead[0] = a.tiers.radius[0]; thead[1] = a.tiers.radius[1]; thead[2] = a.tiers.radius[2]; thead[3] = a.tiers.radius[3];
        thead[4] = a.tiers.radius[4]; thead[5] = a.tiers.radius[5]; thead[6] = a.tiers.radius[6];
        thead[7] = (double)a.tiers.n[0]; thead[8] = (double)a.tiers.n[1]; thead[9] = (double)a.tiers.n[2];
        thead[10] = (double)a.tiers.n[3]; thead[11] = (double)a.tiers.n[4]; thead[12] = (double)a.tiers.n[5];
        thead[13] = (double)a.tiers.n[6];
    }
    const int lane = W > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    // Probe pass of a batched launch: the rows its waves leave alive are gathered per WORKGROUP and reserved in the
    // launch's list with one device-scope atomic (one per wave -- several thousand on one address, ~23 ns each at the
    // memory side -- was what the pass's 65 us were made of)
    constexpr int kWgKeep = (PRUNE && !LONG) ? 8 * kCellsMaxRows : 1;
    __shared__ int wg_keep[kWgKeep];
    __shared__ int wg_nkeep, wg_valid;
    if (PRUNE && !LONG && threadIdx.x == 0) { wg_nkeep = 0; wg_valid = 0x7fffffff; }     // (before the workgroup's only barrier below)
    const bool eblike = (a.model == TRX_MODEL_EB) || (a.model == TRX_MODEL_EB_TWIN);
    const double rs2 = a.rs2;
    const int n_time = a.n_time;
    const float inv_nt = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (float)n_time)));   // (scalar register)
    // the row count: known to the host, or left on the device by an earlier kernel of the stream (the
    // rows per wave and the batch count then follow here, by the host's rule)
    long n = a.n, nbatch = a.nbatch;          // (nbatch: one row per wave only; batches follow batch_plan)
    int B = Bl;
    if (a.n_dev) {
        n = *a.n_dev;
        if (!LONG) {
            B = batch_rows(n, n_time, a.forced_B);
            B = B < Bl ? B : Bl;
        }
        nbatch = (n + B - 1) / B;
    }
#ifdef TRX_PHASE_TIMERS
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    TRX_TICK(t_all);
#endif
    if (!LONG) {
        double* tw = lds_all + a.tl_off;
        for (int j = threadIdx.x; j < n_time; j += 64 * W) {
            tw[j] = a.time[j];
            if (MODE == MODE_LNL) tw[n_time + j] = a.flux[j];
        }
    }
    if (W > 1) __syncthreads();       // the only workgroup barrier: from here on every wave is on its own
    // chi^2 of the flat model (every cell exactly 1): one number per launch (rowc_kernel)
    const double* hdr = a.rowc + n * kRowDoubles;
    // (one number per launch, read from memory: held in a scalar register pair, not in two VGPRs for the whole kernel)
    const double flat_sum = (MODE == MODE_LNL && n_time > 0) ? uniform(hdr[kHdrFlat]) : 0.0;
    if (ST && lane <= 2 * kStM) ss.stw[lane] = hdr[kHdrStW + lane];

    // the rows of this launch: all of them, or (PRUNE) the pilot rows / the rows behind the pilot
    long row0 = 0, row1 = n;
    const int* rlist = nullptr;        // split, part 3: the rows of this launch are rlist[row0 .. row1)
    if (PRUNE && a.part) {
        const long np = n < kPilotRows ? n : kPilotRows;
        // (the argument block's pointers as values first: a choice between a.probe_list and a.surv_list made on the
        // fields themselves is a choice between two addresses INSIDE the block -- see plan_cell)
        const int* const probe_list = a.probe_list;
        const int* const surv_list = a.surv_list;
        const unsigned long long* const probe_count = a.probe_count;
        const unsigned long long* const surv_count = a.surv_count;
        if (a.part == 1) row1 = np;
        else if (a.part == 2) {
            // (split: the probe pass takes the rows depth_screen_kernel listed; when nothing is probed it leaves below)
            if (a.split && hdr[kHdrProbe] != 0.0) { rlist = probe_list; row1 = (long)*probe_count; }
            else row0 = np;
        }
        else if (hdr[kHdrProbe] != 0.0) { rlist = surv_list; row1 = (long)*surv_count; }
        else row0 = np;                // (no probe pass was run: every row behind the pilot, as they come)
        if (!LONG && (a.part == 3 || (a.part == 2 && rlist))) B = listed_pass_rows(a, row1 - row0, a.part);
        // the pilot's rows one per wave: six per wave were 683 waves on 256 CUs for 4096 rows, each a serial chain of a
        // whole batch -- 108 us before the launch proper could start (TOI-465.01, 100 points); short waves fill the chip
        if (!LONG && a.part == 1 && a.split) B = kPilotB < Bl ? kPilotB : Bl;
        nbatch = (row1 - row0 + B - 1) / B;
    }
    // probing pays when many rows lie far above the best (pilot_stats_kernel's verdict; the pilot never probes, nor
    // does the pass over the listed rows)
    // The survivors' pass of the batched variant (part 3) looks at the bound once more: a row survives the probe pass
    // whenever its ~16 probe cells and its out-of-window cells do not prove it negligible -- the unprobed in-window
    // cells are taken to fit perfectly -- and nine survivors in ten are still far from the best (profiles/
    // r03/f_prune_potential.txt: 4.6 % of TOI-465.01's rows survive 16 cells, 0.5 % lie within 90 of the best).  Its first
    // phase takes every a.pstride3-th stamp (a quarter of the row), the verdict drops what that proves negligible, the
    // second phase finishes the rest.
    const bool third_two = PRUNE && !LONG && a.part == 3 && a.pstride3 > 1;
    const int pstride = third_two ? a.pstride3 : a.pstride;
    const bool probing = PRUNE && pstride > 1 && hdr[kHdrProbe] != 0.0 &&
                         (a.part == 0 || a.part == 2 || third_two);
    if (PRUNE && a.split && a.part == 2 && !probing) return;       // nothing to probe: part 3 takes the rows directly
    // (one row per wave: nothing to taper -- an XCD's waves take consecutive rows of its eighth)
    // (a row count read from the device is wave-uniform, which the compiler cannot know: the plan belongs in scalar
    // registers)
    const BatchPlan bp = LONG ? BatchPlan{} : batch_plan(uniform_long(row1 - row0), __builtin_amdgcn_readfirstlane(B), !PRUNE);
    const long positions = LONG ? (nbatch + 7) / 8 : bp.P;
    // workgroups go round the 8 XCDs (blockIdx & 7) and an XCD's waves take consecutive batches of its share
    // of the rows, whose blocks then stay in that XCD's L2
    const long v0 = (long)(blockIdx.x & 7) + 8 * ((long)(blockIdx.x >> 3) * W + wave);
    unsigned n_skipped = 0, n_pruned = 0;          // this wave's rows for trx_skipped_rows / trx_pruned_rows
    for (long v = v0; v < 8 * positions; v += (long)gridDim.x * W) {
        RowsArgs a_again;
        if (kReload == 2) {
            const StarKernArgs* k = kernel_arguments<StarKernArgs>();
            a_again = star_args(k->common, k->bt, k->part);
        }
        const RowsArgs& a = (kReload == 1) ? *kernel_arguments<RowsArgs>() : ((kReload == 2) ? a_again : a_in);
        long base;
        int nb;
        if (LONG) {
            const long batch = (v & 7) * positions + (v >> 3);
            if ((v >> 3) >= positions || batch >= nbatch) continue;
            base = row0 + batch * B;
            nb = (int)((row1 - base < B) ? (row1 - base) : B);
        } else {
            batch_at(bp, v & 7, v >> 3, base, nb);
            base += row0;
            if (base >= row1) continue;
            if (row1 - base < nb) nb = (int)(row1 - base);
        }
        TRX_TICK(t_pro);
        TRX_CENSUS_ADD(kCenBatch, 1);
        // ---- the batch's row blocks (rowc_kernel), one coalesced copy -----------------------
        int rowid = (int)base + lane;      // the row of lane `lane` of the batch (lanes < nb; row counts fit an int: trx_scenario.hip)
        if (PRUNE && rlist) {
            if (lane < nb) rowid = rlist[base + lane];
            double* dst = reinterpret_cast<double*>(rows);
            for (int i = lane; i < nb * kRowDoubles; i += 64) {
                const int r = i / kRowDoubles, q = i - r * kRowDoubles;
                dst[i] = a.rowc[(long)rlist[base + r] * kRowDoubles + q];
            }
        } else {
            const double* src = a.rowc + base * kRowDoubles;
            double* dst = reinterpret_cast<double*>(rows);
            for (int i = lane; i < nb * kRowDoubles; i += 64) dst[i] = src[i];
        }
        wave_sync();
        if (lane < nb) {
            const RowC& c = rows[lane];
            hacc[lane] = 0.0;
            hrem[lane] = 0.0;
            // an unocculted cell: 1 diluted is 1 (or NaN for a degenerate flux ratio)
            double m1 = 1.0;
            m1 = fma(-(1.0 - m1), c.rdil, 1.0);
            hmout[lane] = m1;
        }
        wave_sync();
        // lnL_EB_p returns +inf for a draw whose secondary eclipse is deeper than 1.5 sigma, whatever its
        // light curve looks like (likelihoods.py:535-538): such rows are not evaluated at all
        unsigned long long skipmask = 0;
        if (MODE == MODE_LNL && a.skip_excl && a.model == TRX_MODEL_EB) {
            skipmask = __ballot(lane < nb && rows[lane].excl != 0.0);
            n_skipped += (unsigned)__popcll(skipmask);
            if (LONG && skipmask) {                  // the wave's only row: done
                if (lane == 0) a.out[base] = INFINITY;
                wave_sync();
                continue;
            }
        }
        // LONG: the row constants are wave-uniform -- held in scalar registers they cost no VGPRs
        // and no LDS reads in the pair loop
        // (the window, dilution and exclusion constants are used once per 64 cells: those stay in LDS)
        RowC cu;
        if (LONG && !TRX_LONG_ROWS_IN_LDS) {
            const RowC& r0 = rows[0];
            cu.k = uniform(r0.k); cu.t0 = uniform(r0.t0); cu.nmot = uniform(r0.nmot); cu.e = uniform(r0.e);
            cu.Mtr = uniform(r0.Mtr); cu.ax = uniform(r0.ax); cu.ay = uniform(r0.ay); cu.bx = uniform(r0.bx);
            cu.by = uniform(r0.by); cu.cosi = uniform(r0.cosi); cu.cle = uniform(r0.cle); cu.cld = uniform(r0.cld);
            cu.ced = uniform(r0.ced);
            cu.sEt = uniform(r0.sEt); cu.cEt = uniform(r0.cEt);
        }
        double lacc = 0.0;                 // LONG: this lane's share of the row's chi^2
        bool nonflat = false;              // LONG: a cell of this lane has a model value other than 1
        // PRUNE: lnprior of this lane's row (lanes = rows) and the rows abandoned so far
        unsigned long long deadmask = 0;
        bool long_dead = false;
        if (PRUNE && lane < nb) hlp[lane] = a.prune_lp ? a.prune_lp[row_pos(a, rowid)] : 0.0;
#define lp_row (hlp[lane < nb ? lane : 0])
        bool probe_done = false;           // split, part 2: the batch ends with the verdict
        const unsigned long long exclmask = skipmask;       // the rows the EB secondary rule excludes (+inf)
        // The probe pass (split, part 2) ends with the verdict: an abandoned row reports its bound, an excluded one
        // +inf, and the rows still alive go on the launch's list -- one atomic per wave -- for the pass that evaluates
        // them to the end on full chunks (part 3).  Until round 3 a batch went on with its survivors alone: one or two
        // rows' cells in chunks made for six, behind a second window pass; with nine rows in ten abandoned a call got
        // 15 % faster where the arithmetic allows 2.5 x.
        auto finish_probe = [&]() __attribute__((always_inline)) {
            const bool in_batch = lane < nb;
            const bool is_dead = mask_bit(deadmask, lane), is_excl = mask_bit(exclmask, lane);
            const bool alive = in_batch && !is_dead && !is_excl;
            if (in_batch && !alive) a.out[rowid] = is_excl ? INFINITY : hrem[lane];
            const unsigned long long ma = __ballot(alive);
            if (ma) {
                const int cnt = __popcll(ma);
                int slot = 0;
                if (!LONG) {
                    if (lane == 0) slot = atomicAdd(&wg_nkeep, cnt);
                    slot = __shfl(slot, 0, 64);
                }
                if (!LONG && slot + cnt <= kWgKeep) {
                    if (alive) wg_keep[slot + lanes_below(ma)] = (int)rowid;
                } else {
                    // (the workgroup's buffer is full -- a launch far beyond the grid cap: straight to the list.  Every
                    // later reservation is refused as well: the buffer's entries end where the first refused one began)
                    if (!LONG && lane == 0) atomicMin(&wg_valid, slot);
                    unsigned long long at = 0;
                    if (lane == 0) at = atomicAdd(a.surv_count, (unsigned long long)cnt);
                    at = __shfl(at, 0, 64);
                    if (alive) a.surv_list[at + lanes_below(ma)] = (int)rowid;
                }
            }
        };
#ifndef TRX_NO_DEPTH_SCREEN
        if (PRUNE && probing && !a.split) {
            // depth screen: a row too shallow (diluted) for the data is settled by its constants alone
            // (batches of short light curves: depth_screen_kernel did it before this launch and listed the rest)
            const double hmin_run = __hip_atomic_load(&hdr[kHdrHmin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double xmax_run = __hip_atomic_load(&hdr[kHdrXmax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool shallow = false;
            if (lane < nb && !mask_bit(skipmask, lane)) {
                const double lb = depth_screen(hdr, depth_bound(rows[lane]));
                shallow = hmout[lane] == 1.0 && lb > hmin_run && (a.prune_c0 - lb + lp_row) < xmax_run - 90.0;
#ifdef TRX_PRUNE_NEVER_DEAD
                shallow = false;
#endif
                if (shallow) hrem[lane] = lb;                          // what the row reports
            }
            const unsigned long long ms = __ballot(shallow);
            if (ms) {
                deadmask |= ms;
                skipmask |= ms;
                n_pruned += (unsigned)__popcll(ms);
                wave_sync();
                if (LONG) {                          // the wave's only row
                    if (lane == 0) a.out[base] = hrem[0];
                    wave_sync();
                    continue;
                }
            }
        }
#endif
        const int nphase = probing ? 2 : 1;
        TRX_TOCK(0, t_pro);

        // (every row of the batch settled by the depth screen or the EB rule: no cell to look at)
        const unsigned long long rowsmask = (nb >= 64) ? ~0ull : ((1ull << nb) - 1ull);
        const bool all_settled = PRUNE && !LONG && (skipmask & rowsmask) == rowsmask;
        const int ncell = all_settled ? 0 : nb * n_time;
        // (the probe pass of a split launch files the probe cells only: its window is the whole batch)
        const int wstep = (PRUNE && !LONG && a.part == 2 && a.probe_rows > 1) ? (ncell > kCellsWindow ? ncell : kCellsWindow) : kCellsWindow;
        for (int win0 = 0; win0 < ncell; win0 += wstep) {
            const int win1 = (win0 + wstep < ncell) ? (win0 + wstep) : ncell;
            for (int phase_no = 0; phase_no < nphase; ++phase_no) {
            // pass 1: window test, 64 cells at a time across row boundaries (PRUNE: phase 0 files the probe
            // cells, phase 1 the other cells of the rows still alive)
            TRX_TICK(t_p1);
            int nw = 0;
            for (int c0 = win0; c0 < win1; c0 += 64) {
                const int cell = c0 + lane;
                const bool valid = cell < win1;
                int rr = 0, j = valid ? cell : 0;
                if (!LONG) {
                    rr = valid ? (int)(((float)cell + 0.5f) * inv_nt) : (nb - 1);
                    rr = rr < nb ? rr : nb - 1;
                    j = valid ? (cell - rr * n_time) : 0;
                }
                bool inw = false;
                TRX_CENSUS_ADD(kCenWindowTrip, 1);
                // (a trip all of whose rows are settled -- abandoned, excluded -- has nothing to test)
                if (PRUNE && !LONG && !__any(valid && !mask_bit(skipmask, rr))) continue;
                if (valid) {
                    const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[rr];
                    const double phase = c.nmot * (tl[j] - c.t0);
                    const double dMc = reduce_2pi(phase);
                    const double slack = 1e-15 * fabs(phase);
                    const RowC& cw = rows[rr];                       // (LONG: rr = 0)
                    inw = in_window(cw.wlo - slack, cw.whi + slack, dMc) && !mask_bit(skipmask, rr);
                    // no occultation anywhere in the exposure: the model is 1, diluted
                    if (MODE == MODE_GRID && (!inw || a.debug_nodes))
                        a.out[(size_t)base * n_time + cell] = a.debug_nodes ? 0.0 : hmout[rr];
                    if (LONG && MODE == MODE_LNL && !inw && phase_no == 0) {
                        const double d = fl[j] - 1.0;
                        lacc = fma(d * d, rs2, lacc);                           // :486, :537, :586
                    }
                    if (PRUNE) {
                        // what the flat model charges the row for its in-window cells, until they are done
                        if (!LONG && inw && phase_no == 0) {
                            const double d = fl[j] - 1.0;
                            __hip_atomic_fetch_add(&hrem[rr], (d * d) * rs2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (nphase == 2) {
                            const bool probe = (j % pstride) == (pstride >> 1);
                            inw = inw && (probe == (phase_no == 0));
                        }
                    }
                }
                const unsigned long long mw = __ballot(inw);
                if (inw) winlist[nw + lanes_below(mw)] = (unsigned short)(cell - win0);
                nw += __popcll(mw);
            }
            wave_sync();
            TRX_TOCK(1, t_p1);
            // pass 2: the in-window cells, 64 at a time.  A cell next to a limb contact evaluates
            // all S sub-exposures, the others 3-9 nodes: the first sweep only files those cells
            // (back into the list, behind the read cursor) and a second sweep takes them, so that
            // pairs that may turn out to be off the disc stay out of the first sweep's trips.
            int nheavy = 0;
            for (int sweep = 0; sweep < 2; ++sweep) {
            const int count = sweep ? nheavy : nw;
            // With the centre-value stencil the chunks of the first sweep overlap by kStM cells: a chunk finalises
            // its first 64 - kStM lanes and only lends the centre values of the last kStM, which the next chunk owns;
            // what a chunk's first cells need from their left is carried over from the previous one (see below).
            // (a launch sent here by a stale memo -- the address now holds a light curve without a uniform grid --
            // finds radius 0 and walks the list exactly like the instantiation without the stencil: same chunks,
            // same summation order)
            const bool halo = ST && sweep == 0 && st_radius > 0.0;
            int carry_j = 0;                       // stencil: the time indices of the previous chunk's lanes
            unsigned long long carry_ok = 0;       // ... and which of them were planned cells (no contact cell)
            // Carried cells (no stencil).  A chunk's pairs are dealt to the lanes 64 at a time and the last trip is
            // half empty on average -- 54 of 64 lanes per trip at 100 points, and the pair loop is three quarters of
            // the kernel there.  So a chunk with more cells behind it processes only the cells whose pairs fill whole
            // trips (up to the last cell that ends before the last multiple of 64) and hands the cells behind those --
            // planned, not evaluated -- to the next chunk, where they take the first lanes and the list fills the
            // rest: every trip but a chunk's last is full, and that one lacks at most one cell's pairs.  Cells are
            // still finalised in list order (the carried ones sit in front of the new ones), so a row's chi^2
            // terms are added in the same order as before: same bits.
            // (their plans stay where they are, in the last slots of the cell state, and the next chunk's first lanes read
            // them from there: nothing but the count crosses the loop's back edge -- the batched instantiations have
            // no register to spare)
            int ncarry = 0;                       // cells carried into this chunk: lanes [0, ncarry)
            for (int w0 = 0, step = 64; w0 < count || (kCarry && ncarry > 0); w0 += step) {
                TRX_TICK(t_plan);
                TRX_CENSUS_ADD(sweep ? kCenChunk1 : kCenChunk0, 1);
                bool owned = true;
                const bool carried = kCarry && lane < ncarry;
                if (kCarry) step = 64 - ncarry;
                const int csrc = (lane + 64 - ncarry) & 63;                    // a carried cell's slot in the previous chunk's state
                const unsigned cmeta = carried ? cs.meta[csrc] : 0u;
                bool valid = carried ? (cmeta & 0x20000u) != 0 : (w0 + lane - ncarry) < count;
                int rel = (int)*(carried ? &cs.rel[csrc] : &winlist[valid ? (w0 + lane - ncarry) : (count - 1)]);
                bool last_chunk = false;
                if (ST && halo) {
                    // One-sided halo (round 5): a chunk owns its first 64 - kStM lanes and only lends the centre values
                    // of the last kStM -- the next chunk's first cells.  What its own first cells need from the left --
                    // the centre values, time indices and "planned, not a contact cell" bits of the previous chunk's last
                    // owned cells -- is carried over instead of planning and evaluating those cells a second time
                    // (until then a chunk owned lanes [kStM, 64 - kStM): 52 new cells per plan trip and per 64 centre
                    // pairs; now 58).  The contact cells filed so far are at most the cells owned so far = the next
                    // chunk's first list entry: no entry still to be read is overwritten.
                    last_chunk = w0 + 64 >= count;
                    step = last_chunk ? 64 : 64 - kStM;
                    owned = lane < step;
                    if (w0 > 0 && lane < kStM) ss.fc[lane] = ss.fc[kStM + (64 - 2 * kStM) + lane];
                    wave_sync();
                }
                const int cell = win0 + rel;
                int rr = 0, j = cell;
                if (!LONG) {
                    rr = (int)(((float)cell + 0.5f) * inv_nt);
                    rr = rr < nb ? rr : nb - 1;
                    j = cell - rr * n_time;
                }
                const double t = tl[j];
                double fobs = 0.0;
                if (LONG && MODE == MODE_LNL) fobs = fl[j];          // in flight during the chunk
                CellPlan pl;
                if (carried) {
                    pl.n = (int)(cmeta >> 18); pl.tier = (int)((cmeta >> 8) & 0xffu) - 1;
                    pl.sE = cs.sE[csrc]; pl.cE = cs.cE[csrc];
                    pl.anchored = (cmeta & 0x10000u) != 0;
                } else if (valid) {
                    const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[rr];
#ifndef TRX_PLAN_FULL_SOLVE
                    // (one row per wave only: in the batched variant the second code path costs more -- measured
                    // -2 % at 100-200 points -- than the criteria it skips; +1-3 % here)
                    if (LONG && sweep == 1) pl = plan_all_subexposures<PRUNE>(c, t, a.S);      // filed as such by the first sweep
                    else
#endif
                    pl = plan_cell<false, PRUNE, TRX_TIER_LDS(PRUNE, LONG), kLazyTiers>(c, t, a.exptime, a.S, a.tiers, a.use_tiers != 0,
                                                                                        (ST && sweep == 0) ? st_radius : 0.0, thead);
                    if (STEP && !pl.anchored && pl.n > 0) {
                        // every sub-exposure evaluated (diagnostics): the pairs still step from the centre
                        kepler_full(c.nmot * (t - c.t0) + c.Mtr, c.e, pl.sE, pl.cE);
                        pl.anchored = true;
                    }
                }
                if (sweep == 0 && a.use_tiers) {
                    const bool heavy = valid && !pl.lazy && pl.tier < 0 && pl.n > 0;
                    const unsigned long long mh = __ballot(heavy && owned);       // filed once, by its owner
                    if (heavy && owned) winlist[nheavy + lanes_below(mh)] = (unsigned short)rel;
                    if (heavy) { pl.n = 0; valid = false; }
                    nheavy += __popcll(mh);
                }
                int tier = pl.tier;
                int nodes = (valid && owned) ? pl.n : 0;
                // Centre-value stencil (LONG, dense uniform grid): a cell whose kStM neighbours on either
                // side sit next to it in this chunk, all of them planned cells of this sweep, takes its
                // exposure average from their centre values -- ONE pair, the centre, instead of its
                // Gauss nodes; every cell within kStM of such a cell adds its centre to its own pairs.
                bool st = false, centre = false;
                if (ST && sweep == 0) {
                    // (the kStM cells to the left of this chunk's first: lanes 64 - 2 kStM .. 64 - kStM - 1 of the previous one)
                    // (both shuffles by every lane: a shuffle under a divergent mask reads 0 from the lanes that sit it out)
                    const int jleft = __shfl(carry_j, 64 - 2 * kStM + lane, 64), jhere = __shfl(j, lane - kStM, 64);
                    const int jm = (lane < kStM) ? jleft : jhere;
                    const int jp = __shfl(j, lane + kStM, 64);
                    const unsigned long long mok = __ballot(valid);
                    const unsigned long long left = (carry_ok >> (64 - 2 * kStM)) & ((1ull << kStM) - 1ull);
                    const bool inner = (lane >= kStM || (halo && w0 > 0)) && lane + kStM < 64;
                    constexpr unsigned long long kAll = (1ull << (2 * kStM + 1)) - 1ull;
                    // the 2 kStM + 1 "planned cell" bits around the lane (the lowest ones from the previous chunk)
                    const unsigned long long field = (lane >= kStM) ? (mok >> (lane - kStM)) : (((mok << kStM) | left) >> lane);
                    st = inner && owned && valid && pl.n > 0 && pl.st_ok && jm == j - kStM && jp == j + kStM &&
                         (field & kAll) == kAll;
                    const unsigned long long mst = __ballot(st);
                    unsigned long long dil = mst;
#pragma unroll
                    for (int i = 1; i <= kStM; ++i) dil |= (mst << i) | (mst >> i);
                    centre = valid && ((dil >> lane) & 1ull);
                    // (the next chunk's first cells may take their stencil through this one's last owned cells; one whose
                    // exposure is off the disc altogether has the value 1 without an evaluation)
                    // -- and only a cell that a stencil cell of the next chunk can reach: lane 64 - 2 kStM + k is within
                    // kStM of the next chunk's cells 0 .. k, this chunk's lanes 64 - kStM .. 64 - kStM + k, which can take
                    // the stencil only where their own plan allows it)
                    const unsigned long long mnext = __ballot(valid && pl.n > 0 && pl.st_ok) >> (64 - kStM);
                    const int kk = lane - (64 - 2 * kStM);
                    if (halo && !last_chunk && valid && !centre && kk >= 0 && kk < kStM && (mnext & ((2ull << kk) - 1ull)) != 0) {
                        if (pl.n > 0) centre = true;
                        else ss.fc[kStM + lane] = 1.0;
                    }
                    if (st) nodes = 0;
                    carry_j = j;
                    carry_ok = mok;
                    if (kLazyTiers) {
                        // the cells that need their own nodes after all (no neighbours on one side: a row's first and
                        // last cells in the window): their tier now (plan_cell<LAZY>)
                        const bool need = pl.lazy && nodes > 0;
                        if (__any(need)) {
                            if (need) {
                                const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[rr];
                                plan_tiers(c, a.exptime, pl, thead);
                                tier = pl.tier;
                                nodes = pl.n;
                            }
                        }
                    }
                }
                // A chunk none of whose cells has nodes of its own -- every planned cell takes the stencil or is off the
                // disc: three chunks in five of a row on BASELINE config 1's grid -- evaluates its centres where they
                // were planned, lane = cell, straight from the plan's registers: no pair table, no cell state in LDS, no
                // pair loop (its decode, its second look at the row, its Kepler step).  Same arithmetic on the same
                // numbers: the same centre values, bit for bit.
                const bool direct = kLazyTiers && ST && sweep == 0 && !__any(nodes > 0);
                if (direct) {
                    if (centre) {
                        const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[rr];
                        const double ce = pl.cE - c.e;
                        const double X = fma(c.ax, ce, c.bx * pl.sE);
                        const double Y = fma(c.ay, ce, c.by * pl.sE);
                        const double yc = Y * c.cosi;
                        const double z2 = fma(X, X, yc * yc);
                        const double opp = 1.0 + c.k;
                        double f = 1.0;
                        if (Y >= 0.0 && z2 < opp * opp) {
                            const Limb L{c.cle, c.cld, c.ced};
                            f = disc_flux<FP32, TRX_ATAN_TAB(PRUNE, LONG)>(sqrt_fast(z2), c.k, L, atab);
                        } else if (z2 != z2) {
                            f = z2;
                        }
                        ss.fc[kStM + lane] = f;
                    }
                    wave_sync();
                }
                if (kCarry) wave_sync();        // (the carried cells' reads of the previous chunk's slots are through)
                if (!direct) {
                    cs.sE[lane] = pl.sE; cs.cE[lane] = pl.cE;
                    cs.t[lane] = t;
                    cs.facc[lane] = 0.0;
                    cs.meta[lane] = (unsigned)rr | ((unsigned)(tier + 1) << 8) | (pl.anchored ? 0x10000u : 0u) |
                                    (valid ? 0x20000u : 0u) | ((unsigned)pl.n << 18);
                    if (kCarry) cs.rel[lane] = (unsigned short)rel;
                }
                TRX_TOCK(2, t_plan);
                TRX_TICK(t_a);
                // The (cell, node) pairs of the chunk, cell by cell, dealt to all lanes: a pass
                // takes as many nodes of every cell as fit the pair table (a first-sweep chunk in
                // one pass; 64 contact cells x S = 20 sub-exposures in two).
                int ldone = 64;                    // lanes [0, ldone) are done with this chunk, the others are carried on
                const int ncells = __popcll(__ballot(nodes > 0 || (ST && centre)));
                int per = ncells > 0 ? kCellsPairs / ncells - (ST ? 1 : 0) : kCellsPairs;
                per = per > 1000 ? 1000 : (per < 1 ? 1 : per);
                for (int s0 = 0; !direct && __any(s0 < nodes || (ST && s0 == 0 && centre)); s0 += per) {
                    int cnt = nodes - s0;
                    cnt = cnt < 0 ? 0 : (cnt > per ? per : cnt);
                    TRX_CENSUS_ADD(kCenPass, 1);
                    const int extra = (ST && s0 == 0 && centre) ? 1 : 0;      // the centre rides in the first pass
                    // (first sweep: at most 9 nodes + the centre per cell -- four ballots; contact cells: up to S)
                    int total;
                    const int off = (sweep == 0 && a.use_tiers) ? lane_prefix<4>(cnt + extra, total)
                                                                : lane_prefix<10>(cnt + extra, total);
                    for (int si = 0; si < cnt; ++si) pdesc[off + si] = (unsigned short)(lane | (si << 6));
                    if (extra) pdesc[off + cnt] = (unsigned short)(lane | (kCentreNode << 6));
                    wave_sync();
                    // (carried cells: with more of the list to come and every cell's pairs in this one pass, the cells
                    // whose pairs end within the whole trips are processed, the others handed on -- when the last trip's
                    // empty lanes outnumber the cells that then have to wait)
                    if (kCarry && !(ST && halo) && s0 == 0 && w0 + step < count && total >= 64 && !__any(nodes > per)) {
                        const int whole = total & ~63;
                        const int l0 = __popcll(__ballot(off + cnt + extra <= whole));      // (a prefix of the lanes: off ascends)
                        if (l0 >= 1 && l0 < 64 && (64 - (total - whole)) > (64 - l0)) {
                            ldone = l0;
                            total = __builtin_amdgcn_readlane(off, l0);
                        }
                    }
                    // one pair per lane: the orbit stepped from the cell's centre solution (|dM| <=
                    // half an exposure), the Mandel-Agol flux, and the node's term added to the
                    // cell's sum in LDS (ds_add_f64; a cell's pairs sit in consecutive lanes and the
                    // LDS unit takes them in lane order: node order, bit-repeatable)
                    for (int p0 = 0; p0 < total; p0 += 64) {
                        const int p = p0 + lane;
                        TRX_CENSUS_ADD(kCenPairTrip, 1);
                        TRX_CENSUS_ADD(kCenPairLanes, (unsigned long long)(total - p0 < 64 ? total - p0 : 64));
                        if (p < total) {
                            const int d = (int)pdesc[p];
                            const int h = d & 63;
                            const bool at_centre = ST && (d >> 6) == kCentreNode;
                            const int s = at_centre ? 0 : s0 + (d >> 6);
                            const unsigned meta = cs.meta[h];
                            const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[meta & 0xffu];
                            const int ht = (int)((meta >> 8) & 0xffu) - 1;
                            double sE = cs.sE[h], cE = cs.cE[h];
                            if (!at_centre) {
                                const double frac = (ht < 0) ? fma((double)(s + 1) - 0.5, a.rS, -0.5)
                                                             : tier_xw[2 * (ht * kTierMaxNodes + s)];
                                // mean-anomaly offset of the node from the exposure centre
                                const double dM = c.nmot * (a.exptime * frac);
                                bool have = false;
                                if (STEP && (meta & 0x10000u)) have = kepler_step<PRUNE>(dM, c.e, sE, cE);
                                if (!have) {
                                    TRX_CENSUS_ADD(kCenKeplerFullPair, 1);
                                    kepler_full(c.nmot * ((cs.t[h] + a.exptime * frac) - c.t0) + c.Mtr, c.e, sE, cE);
                                }
                            }
                            const double ce = cE - c.e;
                            const double X = fma(c.ax, ce, c.bx * sE);
                            const double Y = fma(c.ay, ce, c.by * sE);
                            const double yc = Y * c.cosi;
                            const double z2 = fma(X, X, yc * yc);
                            const double opp = 1.0 + c.k;
                            double f = 1.0;
                            if (Y >= 0.0 && z2 < opp * opp) {
                                const Limb L{c.cle, c.cld, c.ced};
                                f = disc_flux<FP32, TRX_ATAN_TAB(PRUNE, LONG)>(sqrt_fast(z2), c.k, L, atab);
                            } else if (z2 != z2) {
                                f = z2;
                            }
                            if (at_centre) {
                                ss.fc[kStM + h] = f;
                            } else {
                                const double term = (ht < 0) ? f : tier_xw[2 * (ht * kTierMaxNodes + s) + 1] * (1.0 - f);
                                if (ht < 0 || term != 0.0)
                                    __hip_atomic_fetch_add(&cs.facc[h], term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    wave_sync();
                }
                TRX_TOCK(3, t_a);
                TRX_TICK(t_rest);
                if (MODE == MODE_GRID && a.debug_nodes) {
                    // bench / test knob: the model evaluations every cell cost, its own and those a
                    // neighbouring chunk spent on its centre value (the cells were zeroed by pass 1)
                    const int spent = nodes + ((ST && centre) ? 1 : 0);
                    if (spent > 0 && lane < ldone) atomicAdd(&a.out[(size_t)base * n_time + cell], (double)spent);
                } else if (valid && owned && lane < ldone) {
                    const RowC& c = (LONG && !TRX_LONG_ROWS_IN_LDS) ? cu : rows[rr];
                    double fsum = cs.facc[lane];
                    if (ST && st) {
                        // the S-point average of the interpolant through the 2 kStM + 1 centre values
                        fsum = 0.0;
#pragma unroll
                        for (int i = -kStM; i <= kStM; ++i) fsum = fma(ss.stw[i + kStM], 1.0 - ss.fc[kStM + lane + i], fsum);
                    }
                    // the cell's flux deficit: nothing, 1 - mean of the S sub-exposures, or the Gauss rule's weighted sum
                    // (a stencil cell's sum is one of deficits like a Gauss rule's -- and its tier may never have been looked for)
                    const double deficit = (pl.n == 0) ? 0.0 : ((ST && st) ? fsum : ((tier < 0) ? 1.0 - fsum / a.dS : fsum));
                    const RowC& cd = rows[rr];
                    const double m = fma(-deficit, cd.rdil, 1.0);    // dilution(s), :352-357, :427-438
                    if (MODE == MODE_GRID) {
                        a.out[(size_t)base * n_time + cell] = m;
                    } else {
                        if (LONG) {
                            // the row's own lanes sum (f - m)^2 / sigma^2 directly    :486, :537, :586
                            const double d = fobs - m;
                            lacc = fma(d * d, rs2, lacc);
                            nonflat = nonflat || (m != 1.0);
                        } else {
                            // (f - m)^2 - (f - 1)^2, exactly 0 for m = 1: one LDS atomic per cell with a
                            // non-unit model -- the wave's lanes meet on 2-3 accumulators and the LDS unit
                            // serialises them in a fixed order, so results repeat bit for bit from run
                            // to run; a six-step shuffle reduction per chunk costs ten times the latency
                            const double f = fl[j];
                            const double contrib = ((1.0 - m) * ((f - m) + (f - 1.0))) * rs2;
                            if (contrib != 0.0)
                                __hip_atomic_fetch_add(&hacc[rr], contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (PRUNE && nphase == 2 && phase_no == 0) {      // this cell's share of chi^2 is now exact
                                const double d1 = f - 1.0;
                                __hip_atomic_fetch_add(&hrem[rr], -(d1 * d1) * rs2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                if (kCarry) ncarry = 64 - ldone;          // the cells handed on (read from this chunk's slots by the next)
                TRX_TOCK(5, t_rest);
            }
            }
            wave_sync();
            if (PRUNE && nphase == 2 && phase_no == 0) {
                // the verdict after the probe cells (and, for free, every out-of-window cell)
                const double* hdr_b = a.rowc + n * kRowDoubles;
                const double hmin_run = __hip_atomic_load(&hdr_b[kHdrHmin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double xmax_run = __hip_atomic_load(&hdr_b[kHdrXmax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (LONG) {
                    double lb = 0.5 * wave_sum(lacc);
                    lb -= fma(1e-9, fabs(lb), 1e-9);                       // summation order
                    if (FP32) lb -= fp32_model_slack(lb, n_time / pstride + 1, rs2);
                    const double lp0 = hlp[0];
                    long_dead = hmout[0] == 1.0 && lb > hmin_run && (a.prune_c0 - lb + lp0) < xmax_run - 90.0;
                    if (long_dead) {
                        if (lane == 0) a.out[base] = lb;
                        ++n_pruned;
                        break;
                    }
                } else {
                    bool dead = false;
                    if (lane < nb && !mask_bit(skipmask, lane)) {
                        double lb = 0.5 * (flat_sum + hacc[lane] - hrem[lane]);
                        lb -= fma(1e-9, fabs(lb) + flat_sum, 1e-9);        // cancellation between the three sums
                        if (FP32) lb -= fp32_model_slack(lb, n_time / pstride + 1, rs2);
                        dead = hmout[lane] == 1.0 && lb > hmin_run && (a.prune_c0 - lb + lp_row) < xmax_run - 90.0;
#ifdef TRX_PRUNE_NEVER_DEAD
                        dead = false;
#endif
                        if (dead) hrem[lane] = lb;                         // what the row reports
                    }
                    const unsigned long long md = __ballot(dead);
                    deadmask |= md;
                    skipmask |= md;
                    n_pruned += (unsigned)__popcll(md);
                    wave_sync();
                    if (a.split && a.part == 2) {
                        finish_probe();
                        probe_done = true;
                    }
                }
            }
            if (PRUNE && probe_done) break;
            }
            if (PRUNE && LONG && long_dead) break;
            if (PRUNE && probe_done) break;
        }
        if (PRUNE && LONG && long_dead) { wave_sync(); continue; }
        if (PRUNE && a.split && a.part == 2 && !probe_done) { finish_probe(); probe_done = true; }     // (all_settled)
        if (PRUNE && probe_done) { wave_sync(); continue; }
        if (MODE == MODE_LNL) {
            if (LONG) {
                // a row whose model is flat over the data takes the launch's flat-model value, so
                // that such rows tie exactly whatever their windows (their lanes would sum the same
                // terms in different orders)
                const double direct = wave_sum(lacc);
                double h = (hmout[0] == 1.0 || n_time == 0) ? 0.5 * (__any(nonflat) ? direct : flat_sum) : NAN;
                if (a.model == TRX_MODEL_EB && rows[0].excl != 0.0) h = INFINITY;   // :535-538
                if (lane == 0) {
                    a.out[base] = h;
                    if (PRUNE && (probing || a.part == 1 || a.part == 3) && h < INFINITY)
                        tighten_bounds(a.rowc + n * kRowDoubles, h, a.prune_c0 - h + lp_row);
                }
            } else {
                double h = INFINITY;
                if (lane < nb) {
                    h = (hmout[lane] == 1.0 || n_time == 0) ? 0.5 * (flat_sum + hacc[lane]) : NAN;
                    if (a.model == TRX_MODEL_EB && rows[lane].excl != 0.0) h = INFINITY;  // :535-538
                    if (PRUNE && mask_bit(deadmask, lane)) h = hrem[lane];
                    a.out[rowid] = h;
                }
                if (PRUNE && (probing || a.part == 1 || a.part == 3)) {
                    // the batch's best finished row tightens the launch's running bounds (one wave, one update)
                    const bool fin = lane < nb && !mask_bit(deadmask, lane) && h < INFINITY;      // (false for NaN)
                    double hb = fin ? h : INFINITY, xb = fin ? a.prune_c0 - h + lp_row : -INFINITY;
                    if (!(xb == xb)) xb = -INFINITY;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        hb = fmin(hb, __shfl_xor(hb, o, 64));
                        xb = fmax(xb, __shfl_xor(xb, o, 64));
                    }
                    if (lane == 0) tighten_bounds(a.rowc + n * kRowDoubles, hb, xb);
                }
            }
        }
        wave_sync();
    }
    if (PRUNE && !LONG && a.split && a.part == 2) {
        __shared__ unsigned long long wg_at;
        __syncthreads();                    // every wave of the workgroup is through its batches
        const int nk = wg_nkeep < wg_valid ? wg_nkeep : wg_valid;
        if (threadIdx.x == 0 && nk > 0) wg_at = atomicAdd(a.surv_count, (unsigned long long)nk);
        __syncthreads();
        for (int i = threadIdx.x; i < nk; i += 64 * W) a.surv_list[wg_at + i] = wg_keep[i];
    }
#undef lp_row
    if (lane == 0) {
        add_row_stat(0, n_skipped);
        add_row_stat(1, n_pruned);
    }
#ifdef TRX_PHASE_TIMERS
    TRX_TOCK(7, t_all);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_cycles[i], tm[i]);
#endif
}

// Whether a launch uses the stencil is decided on the device (rowc_kernel looks at the time stamps),
// so a launch that may use it enqueues BOTH instantiations and the one that does not apply returns
// at once; each gets its own register allocation, and a launch without the stencil runs exactly
// the code it ran before.  The empty launch still costs ~20 us of dispatch, so rowc_kernel leaves
// its verdict in a host-visible memo keyed by the light curve (pointer, length, exposure): later
// launches enqueue only the instantiation it predicts.  A stale memo (the address now holds another
// light curve) is harmless -- the stencil instantiation falls back to the Gauss nodes when the
// device finds no uniform grid, the other one never uses the stencil.
// (Batches: five waves per SIMD for the bounded instantiation too -- until round 4's last day it was compiled for four and
// took 97 VGPRs, one more than five waves allow on 512 registers in granules of 8; for five it takes 95, no scratch:
// the unprobed full evaluations of its third pass gain 8 %, profiles/r04/ab_waves5.txt.)
template <int MODE, bool STEP, bool FP32, bool LONG, bool ST, bool PRUNE = false, int ARGS = 0>
__device__ __forceinline__ void cells_entry(const RowsArgs& a)
{
    // the counter of the secondary-eclipse scan's list (rowc_kernel<true> -> sec_scan_kernel, both done by now) goes
    // back to zero for the next call on this stream
    if (a.need_sec && blockIdx.x == 0 && threadIdx.x == 0) *a.scan_count = 0ull;
    if (a.n_dev) {
        // the grid was sized for an upper bound of the row count: the blocks beyond the batches leave at once
        const long nd = *a.n_dev;
        int B = 1;
        if (!LONG) {
            B = batch_rows(nd, a.n_time, a.forced_B);
            B = B < a.B ? B : a.B;
        }
        long rows_here = nd;
        if (PRUNE && a.part) {
            const long np = nd < kPilotRows ? nd : kPilotRows;
            rows_here = a.part == 1 ? np : nd - np;
            if (!LONG && a.part == 1 && a.split) B = kPilotB < a.B ? kPilotB : a.B;
            if (!LONG && a.part == 2 && a.split) {
                // (the probe pass: the rows depth_screen_kernel listed -- none when nothing is probed)
                rows_here = a.rowc[nd * kRowDoubles + kHdrProbe] != 0.0 ? (long)*a.probe_count : 0;
                B = listed_pass_rows(a, rows_here, 2);
            }
            if (a.part == 3) {
                // (the rows of the third pass: the listed ones, or all behind the pilot; its rows per wave follow from
                // THAT count, as in cells_body)
                if (a.rowc[nd * kRowDoubles + kHdrProbe] != 0.0) rows_here = (long)*a.surv_count;
                // (a.debug_bug, trx_set_debug_bug(1): round 4's first version of this rule, which took the rows per wave from
                // the whole row count when nothing was probed -- the last batches of the third pass were then never
                // written; kept as a switch so that a test can show the "never written" guard catching it)
                if (!LONG && (!a.debug_bug || a.rowc[nd * kRowDoubles + kHdrProbe] != 0.0)) B = listed_pass_rows(a, rows_here, 3);
                else if (!LONG) B = listed_pass_rows(a, nd, 3);             // (the bug: the rule applied to the WHOLE row count)
            }
        }
        // (the first batch index of this workgroup's first wave, see cells_body)
        if (LONG) {
            if ((long)(blockIdx.x & 7) + 8 * ((long)(blockIdx.x >> 3) * cells_waves(LONG)) >= 8 * (((rows_here + B - 1) / B + 7) / 8)) return;
        } else if ((long)(blockIdx.x >> 3) * cells_waves(LONG) >= batch_plan(rows_here, B, !PRUNE).P) return;
    }
    double st_radius = 0.0;
    if (LONG && (ST ? a.use_stencil != 0 : a.use_stencil == 1)) {
        // centre-value stencil of a dense uniform grid: radius in half exposures, 0 = off
        st_radius = uniform(a.rowc[(a.n_dev ? *a.n_dev : a.n) * kRowDoubles + kHdrStRadius]);
        if (a.use_stencil == 1 && (st_radius > 0.0) != ST) return;
        if (!ST) st_radius = 0.0;
    }
    cells_body<MODE, STEP, FP32, LONG, ST, PRUNE, ARGS>(a, st_radius);
}

template <int MODE, bool STEP, bool FP32, bool LONG, bool ST, bool PRUNE = false>
__global__ __launch_bounds__(64 * cells_waves(LONG), LONG ? TRX_CELLS_WAVES_PER_EU : TRX_BATCH_WAVES_PER_EU) void cells_kernel(RowsArgs a)
{
    cells_entry<MODE, STEP, FP32, LONG, ST, PRUNE, 1>(a);
}

// chain (bounded evaluation only): branch = blockIdx.y, `part` = the pass (1 pilot, 2 probe pass / the rest, 3 survivors)
template <bool FP32, bool LONG>
__global__ __launch_bounds__(64 * cells_waves(LONG), LONG ? TRX_CELLS_WAVES_PER_EU : TRX_BATCH_WAVES_PER_EU) void cells_kernel_star(RowsArgs common, BranchTab bt, int part)
{
    const RowsArgs a = star_args(common, bt, part);
    cells_entry<MODE_LNL, true, FP32, LONG, false, true, 2>(a);
}

// ---- the small kernels between the passes of the bounded evaluation ----------------------------------------
// After the pilot launch: does probing pay?  One workgroup over the pilot rows' chi^2/2: if fewer than 90 %
// of the finite ones lie more than 150 above the smallest, the main launch evaluates its rows in one pass
// (a scenario no draw of which comes near the data -- a faint neighbour that would need a 50 % deep eclipse --
// has all its rows within a few tens of each other: nothing to abandon; and measured per call in round 4,
// profiles/r04/bounded_short.txt: with 37 % of the rows abandoned -- TOI-411.02, a 166 ppm signal -- the probe pass
// costs more than it saves (0.54 -> 0.73 ms), with 76 % it pays (0.59 -> 0.50), with 93 % it halves the call.  The share
// of pilot rows 150 above the best overstates what the probe cells can prove: 0.8 for TOI-411.02, 0.995 and more for
// the cases that gain -- hence 90 %).
__device__ __forceinline__ void pilot_stats_body(const double* __restrict__ h, long n, const long* __restrict__ n_dev,
                                                 double* __restrict__ rowc, unsigned long long* __restrict__ surv_count,
                                                 int pstride, unsigned long long* __restrict__ probe_count)
{
    if (threadIdx.x == 0 && surv_count) *surv_count = 0ull;
    if (threadIdx.x == 0 && probe_count) *probe_count = 0ull;
    __shared__ double smin[4];
    __shared__ int sfar[4], sfin[4];
    if (n_dev) n = *n_dev;
    const long np = n < kPilotRows ? n : kPilotRows;
    double m = INFINITY;
    for (long i = threadIdx.x; i < np; i += 256) {
        const double v = h[i];
        if (v < m) m = v;                               // (false for NaN)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    int far = 0, fin = 0;
    for (long i = threadIdx.x; i < np; i += 256) {
        const double v = h[i];
        fin += (v < INFINITY) ? 1 : 0;
        far += (v < INFINITY && v > m + 150.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { far += __shfl_xor(far, o, 64); fin += __shfl_xor(fin, o, 64); }
    if ((threadIdx.x & 63) == 0) { sfar[threadIdx.x >> 6] = far; sfin[threadIdx.x >> 6] = fin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        far = sfar[0] + sfar[1] + sfar[2] + sfar[3];
        fin = sfin[0] + sfin[1] + sfin[2] + sfin[3];
        // (the same condition as cells_body's `probing`: a stride of 1 leaves nothing to probe)
        rowc[n * kRowDoubles + kHdrProbe] = (pstride > 1 && fin > 0 && 100L * far >= 90L * fin) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void pilot_stats_kernel(const double* __restrict__ h, long n, const long* __restrict__ n_dev,
                                                          double* __restrict__ rowc, unsigned long long* __restrict__ surv_count,
                                                          int pstride, unsigned long long* __restrict__ probe_count)
{
    pilot_stats_body(h, n, n_dev, rowc, surv_count, pstride, probe_count);
}

__global__ __launch_bounds__(256) void pilot_stats_kernel_star(RowsArgs common, BranchTab bt)
{
    const RowsArgs a = star_args(common, bt, 0);
    pilot_stats_body(a.out, a.n, a.n_dev, a.rowc, a.surv_count, a.pstride, a.probe_count);
}

// Depth screen of the rows behind the pilot (bounded evaluation of batches; after pilot_stats_kernel, before the probe
// pass): lanes = rows.  A row too shallow (diluted) for the data is settled by its constants alone and reports the
// bound; the others go on the probe pass's list, one atomic per wave.  Until round 4's last day the probe pass screened
// its own rows: with four rows in five settled (TOI-465.01's lnZ_TTP: 82 %) its batches held one live row of six, and
// the window pass -- 64 cells at a time over ALL cells of a batch -- ran on a fifth of its lanes.
constexpr int kScreenRows = 512;           // rows per workgroup of depth_screen_kernel (two trips per wave)
__device__ __forceinline__ void depth_screen_body(const RowsArgs& a)
{
    // One reservation in the probe pass's list per WORKGROUP (512 rows): a device-scope atomic on one address costs
    // ~23 ns at the memory side whoever issues it, and one per wave of 64 rows -- 1600 of them for 10^5 rows -- made
    // this kernel 25 us long.  The depth table goes through LDS (depth_screen reads it three times in a row).
    __shared__ double tab[2 * 64];
    __shared__ int keep[kScreenRows];
    __shared__ int nkeep;
    __shared__ unsigned long long base_at;
    const long n = a.n_dev ? *a.n_dev : a.n;
    const double* hdr = a.rowc + n * kRowDoubles;
    if (hdr[kHdrProbe] == 0.0) return;                 // nothing is probed: the third pass takes the rows as they come
    const long np = n < kPilotRows ? n : kPilotRows;
    const int lane = (int)(threadIdx.x & 63);
    if (threadIdx.x < 128) tab[threadIdx.x] = hdr[kHdrGrid + threadIdx.x];
    const double hmin = hdr[kHdrHmin], xmax = hdr[kHdrXmax];
    const bool excl_rule = a.skip_excl && a.model == TRX_MODEL_EB;
    unsigned n_pruned = 0;
    for (long w0 = np + (long)blockIdx.x * kScreenRows; w0 < n; w0 += (long)gridDim.x * kScreenRows) {
        if (threadIdx.x == 0) nkeep = 0;
        __syncthreads();
        for (long r0 = w0 + (threadIdx.x & ~63); r0 < w0 + kScreenRows && r0 < n; r0 += 256) {
            const long row = r0 + lane;
            bool alive = row < n, shallow = false;
            double lb = 0.0;
            if (alive) {
                const RowC& c = *reinterpret_cast<const RowC*>(a.rowc + row * kRowDoubles);
                // (a row the EB secondary rule excludes stays on the list: the probe pass reports its +inf and counts it)
                if (!(excl_rule && c.excl != 0.0)) {
                    lb = depth_screen(tab - kHdrGrid, depth_bound(c));
                    const double lp = a.prune_lp ? a.prune_lp[row_pos(a, row)] : 0.0;
                    // (an unocculted cell must read exactly 1: a degenerate flux ratio makes it NaN, and so the row's chi^2)
                    const double m1 = fma(-(1.0 - 1.0), c.rdil, 1.0);
                    shallow = m1 == 1.0 && lb > hmin && (a.prune_c0 - lb + lp) < xmax - 90.0;
#if defined(TRX_PRUNE_NEVER_DEAD) || defined(TRX_NO_DEPTH_SCREEN)
                    shallow = false;
#endif
                }
                if (shallow) a.out[row] = lb;
                alive = !shallow;
            }
            n_pruned += (unsigned)__popcll(__ballot(shallow));
            const unsigned long long ma = __ballot(alive);
            if (ma) {
                int at = 0;
                if (lane == 0) at = atomicAdd(&nkeep, __popcll(ma));
                at = __shfl(at, 0, 64);
                if (alive) keep[at + lanes_below(ma)] = (int)row;
            }
        }
        __syncthreads();
        const int nk = nkeep;
        if (threadIdx.x == 0 && nk) base_at = atomicAdd(a.probe_count, (unsigned long long)nk);
        __syncthreads();
        for (int i = threadIdx.x; i < nk; i += 256) a.probe_list[base_at + i] = keep[i];
        __syncthreads();
    }
    if (lane == 0 && n_pruned) add_row_stat(1, n_pruned);
}

__global__ __launch_bounds__(256) void depth_screen_kernel(RowsArgs a)
{
    depth_screen_body(a);
}

__global__ __launch_bounds__(256) void depth_screen_kernel_star(RowsArgs common, BranchTab bt)
{
    const RowsArgs a = star_args(common, bt, 0);
    depth_screen_body(a);
}

}  // namespace
