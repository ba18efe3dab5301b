// Internals shared by the translation units of libtrx.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/trx.h"

namespace trx {

struct ScenFinal;      // trx_device.hpp

// Per-(device, stream) scratch owned by the library.  Work on one stream is ordered, so a buffer can
// serve call after call on that stream without any allocator traffic; it only grows (the stream is
// synchronised first, then hipFree + hipMalloc), and it is released by trx_release_scratch().
//   slot 0  row-constant blocks of a likelihood call (rowc_kernel -> cells_kernel)
//   slot 1  trx_scenario_evidence: buffers sized by the number of draws
//   slot 2  trx_star_enqueue: the arena of a launch chain (several calls' buffers side by side)
//   slot 3  (host, pinned) small results on their way back
constexpr int kScratchSlots = 4;
// The first kScratchZeroed bytes of a device slot are zero when the buffer is handed out for the first time and
// after every growth (cleared on the stream): persistent counters that the kernels themselves leave at zero.
constexpr size_t kScratchZeroed = 4096;
hipError_t stream_scratch(hipStream_t st, int slot, size_t bytes, void** out);
// Scratch of a call that is being captured into a hipGraph on `st`: a buffer of its own, owned by the graph under
// capture and handed back to a pool when that graph and its executables are gone (see trx_kernels.hip).
hipError_t capture_scratch(hipStream_t st, size_t bytes, void** out);
void capture_scratch_stats(long* live, long* idle);      // buffers graphs still own / buffers waiting for reuse
hipError_t capture_scratch_release_idle();

// Held while a call enqueues its kernels on `st`: two host threads that share a stream take turns, so
// the kernels of one call (rowc_kernel -> cells_kernel -> reductions, all on the stream's scratch) are
// enqueued back to back and the stream's order does the rest.  Re-entrant on the owning thread.
class StreamLock {
public:
    explicit StreamLock(hipStream_t st);
    ~StreamLock();
    StreamLock(const StreamLock&) = delete;
    StreamLock& operator=(const StreamLock&) = delete;

private:
    void* mu_;
};

// The likelihood of the draws that passed a geometry mask.  compact_fill_kernel stored their columns DENSELY in the
// [n_param][src_stride] block, in list order: row r at column position r, the twin branch's rows from the top down
// (src_stride - 1 - r); r < *n_dev (left on the device by earlier kernels of the stream; n_upper bounds the count).
// (Until round 5 the columns sat at their draw index and were read through src_idx: one 64-byte line per column and
// row, 700 MB of traffic per target in rowc_kernel alone.)  twin: the EB_TWIN rows (2 P, a of 2 P).
// The caller reduces the rows to lnZ (with lnprior per masked draw, stored like the columns, or null) and the best draw
// only, which allows the bounded evaluation of cells_kernel<PRUNE>: a row that can neither carry weight nor be the
// best draw reports a lower bound of its chi^2/2 instead of the value.  *bounds_base then points at the row
// blocks of the launch, behind which its header keeps the largest log-weight (lme_draws reads it; the
// pointer stays valid for work enqueued on this stream); null when every row was evaluated to the end.
int lnl_draws(int model, int flags, const double* time, const double* flux, int n_time, double sigma,
              const double* cols, long n_upper, const long* n_dev, const int* src_idx, long src_stride,
              int twin, double exptime, int nsupersample, double* out_halfchi2, const double* lnprior,
              double lnsigma, const double** bounds_base, hipStream_t st);

// First pass of the evidence and of the best-draw search over those chi^2/2 values: per-block
// (max, sum exp, saw +inf) partials in ws[3 * 2048] and (value, position) argmin partials in
// amin_pv / amin_pi [2048]; lme_blocks(*n_dev) of them are valid.
// fin.state != null: the block that finishes last also folds the partials into the branch's record
// (scenario_final, trx_device.hpp).
int lme_draws(const double* halfchi2, const double* lnprior, double lnsigma, long n_upper, const long* n_dev,
              const int* src_idx, double* ws, double* amin_pv, long* amin_pi, const double* bounds_base,
              const ScenFinal& fin, hipStream_t st);

// ---- one launch chain for several lnZ_* calls (trx_star_enqueue) -------------------------------------------------
// A branch of a chain: what lnl_draws + lme_draws take for one branch of one call, with the branch's own scratch
// (chain_branch_scratch_bytes(N) bytes; `scan_count` = 8 bytes that are zero before the chain's first use of them and
// at a place nothing else ever occupies) and reduction buffers (ws [3 * 2048], amin_pv [2048], amin_pi [2 * 2048]).
struct ChainBranch {
    int model, flags, twin;
    const double* flux;
    double sigma, lnsigma;
    const double* cols;            // the call's [ncol][N] column block
    const long* n_dev;
    const int* src_idx;
    double* h;
    const double* lnprior;         // per draw, or null
    double* scratch;
    unsigned long long* scan_count;
    double* ws;
    double* amin_pv;
    long* amin_pi;
    const ScenFinal* fin;
};
constexpr int kChainNotApplicable = -1;
// the flags every call of a chain must share (the launch plan is made from the first call's)
constexpr int kChainSharedFlags = TRX_FLAG_FP32_MODEL | TRX_FLAG_EVALUATE_EXCLUDED | TRX_FLAG_ALL_SUBEXPOSURES |
                                  TRX_FLAG_NO_STENCIL | TRX_FLAG_FULL_EVALUATION;
constexpr int kChainMaxCalls = 16, kChainMaxBranchesHost = 24;
size_t chain_branch_scratch_bytes(long n_upper);
// would lnl_lme_chain take these rows?  (the bounded evaluation's passes apply: see lnl_lme_chain)
bool lnl_chain_applicable(int flags, int n_time, long N, int S);
// A block of pinned host memory for a small upload enqueued on `st` right away: begin -> fill the block ->
// hipMemcpyAsync(..., st) -> end (records an event: the block is handed out again only after the copy has run).
// A ring of blocks per (device, stream); begin waits for the oldest copy when all are in flight.
hipError_t pinned_stage_begin(hipStream_t st, size_t bytes, void** block, void** ticket);
void pinned_stage_end(hipStream_t st, void* ticket);
int lnl_lme_chain(const ChainBranch* br, int nbr, const double* time, int n_time, long N, double exptime, int S,
                  hipStream_t st);
// the draw kernels of a chain's calls, the call as the grid's second (draw) / third (fill) dimension: `tab` = the calls'
// argument blocks in DEVICE memory (output pointers set), blk_cnt [n_calls][2 * kDrawMaxGroups]
struct ChainFill {
    int* idx0;
    int* idx1;
    long* n_dev;
    double* cols0;         // [ncol]: draw 0's columns (the stand-in for the best draw of a branch no draw passed)
};
// (trx_scenario_enqueue with a table of K best draws: cols0 is [ncol][n_pad], the columns of draws 0 .. n_pad - 1)
int draw_chain(const trx_draw_args* host_args, const trx_draw_args* dev_tab, int n_calls, int* blk_cnt, const ChainFill* fills,
               long* per_out, int* groups_out, hipStream_t st);

// trx_draw_scenario with the first half of the ordered compaction: workgroup b takes the draws
// [b * per, (b + 1) * per) and leaves its mask counts in blk_cnt[b] / blk_cnt[groups + b] (twin branch)
// (1024 workgroups of 256 threads for N = 1e6, ~1000 draws each: of a chunk of 1024 pre-tested draws ~100 go on to the
// fp64 mask, on 256 lanes; with 2048 workgroups of ~500 draws the fp64 pass ran on a fifth of its lanes -- draw_kernel<2>
// 43.5 -> 39.8 us per call, compact_fill_kernel 33.6 -> 38.8 (half as many one-wave workgroups), 64-TOI step 0.173 ->
// 0.166 s, 75 scenarios 19.0 -> 18.3 ms: profiles/r04/ab_draw.txt; 512 workgroups lose)
#ifndef TRX_DRAW_GROUPS
#define TRX_DRAW_GROUPS 1024
#endif
constexpr int kDrawMaxGroups = TRX_DRAW_GROUPS;
int draw_counted(const trx_draw_args& a, int* blk_cnt, long* per_out, int* groups_out, hipStream_t st);
// behind it: the ordered lists of the draws that passed a mask (idx0 / idx1, their lengths in n_dev[0 / 1]) and the
// columns and the prior of those draws (and of draw 0)
int compact_fill(const trx_draw_args& a, long per, int groups, const int* blk_cnt, int* idx0, int* idx1, long* n_dev,
                 double* cols0, hipStream_t st, int n_pad = 1);

// fail() of trx_kernels.hip for the other translation units (thread-local message of trx_last_error)
int fail_hip(hipError_t e);

}  // namespace trx
