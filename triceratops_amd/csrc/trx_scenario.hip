// trx_scenario_evidence / trx_scenario_enqueue / trx_star_enqueue (include/trx.h): one lnZ_* call of calc_probs, end
// to end on the device, with no host synchronisation inside the call.
//
// Round 2's version strung the call together from 12 (planet) or 23 (binary) launches and synchronised the stream
// twice: the number of draws that pass the geometry mask sized the likelihood launch, so the host had to read it.
// Here the count never leaves the device:
//   draw_kernel          the geometry mask(s) of every draw (trx_draw.hip: what does not feed a mask is not computed,
//                        no column is written); workgroup b takes draws [b per, (b + 1) per) and leaves its mask counts
//   compact_fill_kernel  ordered compaction of the mask(s) -- a workgroup sums the counts of the workgroups before it
//                        and appends the indices of its own masked draws, ascending (the order numpy's / torch's
//                        nonzero gives) -> idx[branch][], n[branch] -- and, in the same kernel, the parameter columns
//                        and the prior of those draws (the 5-10 % that passed), recomputed from the same counter-based
//                        random numbers and stored DENSELY, in list order (branch 0 from position 0 up, the twin branch
//                        from position N - 1 down: a draw passes at most one of the two masks)
//   rowc_kernel          } the likelihood of the masked draws: coalesced reads of those dense columns; the row count
//   (sec_scan_kernel)    } is read from n[branch] on the device and the grids are sized for a guess
//   cells_kernel         } (trx_cells.hpp)
//   lme_partial_kernel   first pass of the log-mean-exp and of the search for the smallest chi^2 in one pass; the block
//                        that finishes LAST folds the partials into the record: the evidence, the best draw (first of
//                        equals, NaN first: numpy's / torch's argmin), its columns, the masked count and the
//                        limb-darkening flag (scenario_final, trx_device.hpp), written straight into the caller's
//                        pinned record
// (with the bounded evaluation cells_kernel is up to five launches: pilot rows, pilot_stats_kernel, depth_screen_kernel,
// probe pass, the rows left alive -- trx_cells.hpp)
// 5 launches for a planet scenario (7 in round 3), 9 for a binary one (two branches; 10), no memset, no copy, NO
// sync: a caller can enqueue every lnZ_* call of a calc_probs on a few streams and wait once (trx_scenario_enqueue);
// trx_scenario_evidence is the same followed by one hipStreamSynchronize.  Every buffer lives in the stream's scratch
// (trx_internal.hpp), behind a small persistent block (the finished-block counter of the last stage, the draw
// kernel's flag) that the kernels leave at zero.  Results are bit for bit those of the torch-operator chain of
// fused.py (same arithmetic on the same rows in the same order).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../include/trx.h"
#include "trx_device.hpp"
#include "trx_internal.hpp"
#include "trx_knobs.hpp"

namespace {

#define TRXS_HIP(call)                                   \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return trx::fail_hip(e_);  \
    } while (0)

// bump allocator over one of the stream's scratch buffers (trx_internal.hpp): sizes first, then pointers
struct Arena {
    size_t used = 0;
    char* base = nullptr;
    size_t reserve(size_t bytes)
    {
        const size_t at = used;
        used += (bytes + 255) & ~(size_t)255;
        return at;
    }
    template <class T>
    T* at(size_t off) const { return reinterpret_cast<T*>(base + off); }
};

constexpr int kLmeParts = 2048;            // lme_blocks() never exceeds it

// ---------------------------------------------------------------------------------------------------------------
// The reference's table of the K best draws of a branch (marginal_likelihoods.py:152-171): the masked draws with the K
// smallest chi^2 (NaN last; exact ties: the earlier draw first), in that order, and the columns gathered at them.
// Not a hot path -- calc_probs keeps the best draw only, which the reduction's final stage already finds -- but what a
// caller of lnZ_* itself gets back, and until round 6 only the torch-operator chain could produce it (torch.topk).
// ONE workgroup streams the branch's chi^2 values through LDS, 2048 - 128 new ones at a time next to the 128 best so
// far, and sorts the 2048 (bitonic, (value, row) keys) whenever a new one can enter the best 128; then thread j
// gathers row j's columns -- or, past the masked count, the columns of draw j - count from the stand-in block
// compact_fill_kernel filled (the reference's table then holds draws of log-likelihood -inf in argsort's order of
// equals; the operator chain pads with draws 0, 1, 2 ...: so does this).
constexpr int kTableKeep = 128, kTableTile = 2048;
static_assert(TRX_TABLE_MAX_ROWS + 1 <= kTableKeep, "one more than the table's rows is kept (the caller's tie check)");
struct TableArgs {
    const double* h;          // chi^2/2 of the masked draws, in list order
    const long* n_dev;        // their number
    const double* cols;       // [ncol][N], dense (row r at position r; the twin branch: N - 1 - r)
    const double* cols_pad;   // [ncol][K]: draws 0 .. K - 1
    long N;
    int ncol, branch, K;
    double* table;            // this branch's TRX_TABLE_BRANCH(K) doubles
};

__device__ __forceinline__ bool table_before(double a, int ia, double b, int ib)
{
    const bool na = a != a, nb = b != b;
    if (na != nb) return nb;                 // NaN last (numpy's argsort, torch.topk(largest=False))
    if (!na && a != b) return a < b;
    return ia < ib;
}

__global__ __launch_bounds__(256) void table_kernel(TableArgs t)
{
    __shared__ double val[kTableTile];
    __shared__ int row[kTableTile];
    __shared__ int any_new;
    const long n = *t.n_dev;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kTableKeep; i += 256) { val[i] = NAN; row[i] = 0x7fffffff; }      // (sorts behind every real entry)
    __syncthreads();
    for (long base = 0; base < n || base == 0; base += kTableTile - kTableKeep) {
        if (tid == 0) any_new = 0;
        __syncthreads();
        const double worst = val[kTableKeep - 1];
        const int wrow = row[kTableKeep - 1];
        bool mine = false;
        for (int i = kTableKeep + tid; i < kTableTile; i += 256) {
            const long r = base + (i - kTableKeep);
            const bool in = r < n;
            const double v = in ? t.h[r] : NAN;
            val[i] = v;
            row[i] = in ? (int)r : 0x7fffffff;
            mine = mine || (in && table_before(v, (int)r, worst, wrow));
        }
        if (mine) any_new = 1;
        __syncthreads();
        if (any_new) {
            for (int k = 2; k <= kTableTile; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < kTableTile; i += 256) {
                        const int p = i ^ j;
                        if (p > i) {
                            const bool up = (i & k) == 0;
                            const double a = val[i], b = val[p];
                            const int ia = row[i], ib = row[p];
                            if (table_before(b, ib, a, ia) == up) { val[i] = b; row[i] = ib; val[p] = a; row[p] = ia; }
                        }
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (n == 0) break;
    }
    const int K = t.K;
    if (tid <= K) {
        const bool real = row[tid] != 0x7fffffff;
        // the K + 1 smallest values (NaN where there is no such draw)
        t.table[14 * (K + 1) + tid] = real ? val[tid] : NAN;
        if (tid < K) {
            const long r = row[tid];
            const long pos = t.branch ? t.N - 1 - r : r;
            const long pad = ((long)tid - n) % (t.N > 0 ? t.N : 1);           // draw index of a spare row (tid >= n there)
            for (int c = 0; c < t.ncol; ++c)
                t.table[c * (K + 1) + tid] = real ? t.cols[(long)c * t.N + pos] : t.cols_pad[(long)c * K + (pad < K ? pad : 0)];
        }
    }
}

int enqueue(const trx_scenario_args* s, double* out_host, hipStream_t st)
{
    trx_draw_args d = *s->draw;
    const long N = d.N;
    if (N < 1 || N > 0x7fffffffL) return TRX_ERR_ARG;
    const int planet = d.planet != 0;
    const int ncol = planet ? 11 : 14, nbr = planet ? 1 : 2;
    // a table of the K best draws (include/trx.h): every masked draw evaluated to the end, K stand-in draws
    const int K = s->table_rows > 1 ? s->table_rows : 0;
    if (K > TRX_TABLE_MAX_ROWS || (K && !s->table)) return TRX_ERR_ARG;
    const int n_pad = K ? K : 1;
    const int flags = s->flags | (K ? TRX_FLAG_FULL_EVALUATION : 0);
    trx::StreamLock turn(st);              // the whole call is enqueued back to back on the stream's scratch

    // Where the record goes: straight into the caller's buffer when the device can write there (pinned host memory:
    // hipHostMalloc / torch pin_memory are mapped into the device's address space), else through a device copy
    double* rec_dev = nullptr;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, out_host) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
            rec_dev = static_cast<double*>(attr.devicePointer);
        else
            (void)hipGetLastError();
    }

    Arena A;
    const size_t o_state = A.reserve(trx::kScratchZeroed);     // persistent: finished-block counter, the draw kernel's flag
    const size_t o_cols = A.reserve(sizeof(double) * ncol * N), o_mask = A.reserve(N), o_mask2 = A.reserve(planet ? 0 : N),
                 o_prior = A.reserve(s->want_prior ? sizeof(double) * N : 0),
                 o_n = A.reserve(2 * sizeof(long)), o_cols0 = A.reserve(sizeof(double) * 16 * n_pad),
                 o_table = A.reserve(K ? sizeof(double) * 2 * TRX_TABLE_BRANCH(K) : 0),
                 o_res = A.reserve(sizeof(double) * (2 * TRX_SCENARIO_OUT + 1)),
                 o_ws = A.reserve(sizeof(double) * 2 * 3 * kLmeParts), o_pv = A.reserve(sizeof(double) * 2 * kLmeParts),
                 o_pi = A.reserve(sizeof(long) * 2 * 2 * kLmeParts), o_cnt = A.reserve(sizeof(int) * 2 * trx::kDrawMaxGroups),
                 o_idx0 = A.reserve(sizeof(int) * N), o_idx1 = A.reserve(planet ? 0 : sizeof(int) * N),
                 o_h0 = A.reserve(sizeof(double) * N), o_h1 = A.reserve(planet ? 0 : sizeof(double) * N);
    TRXS_HIP(trx::stream_scratch(st, 1, A.used, reinterpret_cast<void**>(&A.base)));
    unsigned* state = A.at<unsigned>(o_state);
    d.cols = A.at<double>(o_cols);
    d.mask = A.at<unsigned char>(o_mask);
    d.mask_twin = planet ? nullptr : A.at<unsigned char>(o_mask2);
    d.lnprior = s->want_prior ? A.at<double>(o_prior) : nullptr;
    d.flag = reinterpret_cast<int*>(state + 1);        // zero between calls: the last branch's final stage clears it
    d.dump = nullptr;
    long* n_dev = A.at<long>(o_n);
    double* res = rec_dev ? rec_dev : A.at<double>(o_res);
    int* idx[2] = {A.at<int>(o_idx0), planet ? nullptr : A.at<int>(o_idx1)};
    double* h[2] = {A.at<double>(o_h0), planet ? nullptr : A.at<double>(o_h1)};
    long per = 0;
    int groups = 0;
    if (int rc = trx::draw_counted(d, A.at<int>(o_cnt), &per, &groups, st)) return rc;
    // from here on a failure leaves kernels enqueued that may have touched the persistent block: cleared on the way out
    auto bail = [&](int rc) {
        (void)hipMemsetAsync(state, 0, trx::kScratchZeroed, st);
        return rc;
    };
    if (int rc = trx::compact_fill(d, per, groups, A.at<int>(o_cnt), idx[0], idx[1], n_dev, A.at<double>(o_cols0), st, n_pad))
        return bail(rc);
    double* table_dev = nullptr;           // where the device writes the table: the caller's buffer if it can
    if (K) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, s->table) == hipSuccess && attr.devicePointer &&
            (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice))
            table_dev = static_cast<double*>(attr.devicePointer);
        else
            (void)hipGetLastError();
    }
    if (trx::knob_poison())          // tests: an unwritten row must show (include/trx_debug.h)
        for (int b = 0; b < nbr; ++b) TRXS_HIP(hipMemsetAsync(h[b], 0, sizeof(double) * (size_t)N, st));
    for (int b = 0; b < nbr; ++b) {
        const int model = planet ? TRX_MODEL_TP : (b ? TRX_MODEL_EB_TWIN : TRX_MODEL_EB);
        const double* bounds = nullptr;
        if (int rc = trx::lnl_draws(model, flags, s->time, s->flux, s->n_time, s->sigma, d.cols, N, n_dev + b, idx[b], N,
                                    b, s->exptime, s->nsupersample, h[b], d.lnprior, s->lnsigma, &bounds, st))
            return bail(rc);
        trx::ScenFinal fin{};
        fin.idx = idx[b];
        fin.cols = d.cols;
        fin.cols0 = A.at<double>(o_cols0);
        fin.cols0_stride = n_pad;
        fin.dense = 1;
        fin.N = N;
        fin.n_total = N;
        fin.ncol = ncol;
        fin.branch = b;
        fin.last_branch = (b == nbr - 1) ? 1 : 0;
        fin.res = res + (size_t)b * TRX_SCENARIO_OUT;
        fin.flag_out = (b == 0) ? res + 2 * TRX_SCENARIO_OUT : nullptr;
        fin.state = state;
        if (int rc = trx::lme_draws(h[b], d.lnprior, s->lnsigma, N, n_dev + b, idx[b],
                                    A.at<double>(o_ws) + (size_t)b * 3 * kLmeParts, A.at<double>(o_pv) + (size_t)b * kLmeParts,
                                    A.at<long>(o_pi) + (size_t)b * 2 * kLmeParts, bounds, fin, st))
            return bail(rc);
        if (K) {
            TableArgs t{};
            t.h = h[b]; t.n_dev = n_dev + b; t.cols = d.cols; t.cols_pad = A.at<double>(o_cols0); t.N = N;
            t.ncol = ncol; t.branch = b; t.K = K;
            t.table = (table_dev ? table_dev : A.at<double>(o_table)) + (size_t)b * TRX_TABLE_BRANCH(K);
            hipLaunchKernelGGL(table_kernel, dim3(1), dim3(256), 0, st, t);
            if (hipGetLastError() != hipSuccess) return bail(TRX_ERR_HIP);
        }
    }
    if (K && !table_dev)
        TRXS_HIP(hipMemcpyAsync(s->table, A.at<double>(o_table), sizeof(double) * (size_t)nbr * TRX_TABLE_BRANCH(K),
                                hipMemcpyDefault, st));
    if (!rec_dev)
        TRXS_HIP(hipMemcpyAsync(out_host, res, sizeof(double) * (2 * TRX_SCENARIO_OUT + 1), hipMemcpyDeviceToHost, st));
    return TRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One launch chain for several calls: draw_kernel, compact_fill_kernel, rowc_kernel, (sec_scan_kernel,) the passes of the
// bounded evaluation and lme_partial_kernel are launched ONCE each, with the call (draw side) or the branch (likelihood
// side) as a further grid dimension -- 11 launches and one small upload for up to 16 calls / 24 branches, where the calls
// one by one take 9 (planet) to 17 (binary) launches EACH.  Round 4 measured why that matters (profiles/
// r04/concurrency_levels.txt, r04/j_batch_kernel_stats.txt): the chip runs ~3 kernels at a time whatever the number of
// streams, a 64-target step was 10 183 launches, most of them 8-45 us long on a tenth of the machine, and the host spent
// 113 of the step's 155 ms enqueueing them.  The calls of a chain must share N, the time stamps, the exposure settings
// and the precision flag (the calls of one target do: triceratops.py:767-1428); flux and sigma may differ (a nearby
// star's light curve is the target's, renormalised).  Results are those of the calls one by one, bit for bit: the same
// kernels' bodies on the same rows in the same order.
constexpr size_t kBranchHead = 64;       // bytes of the zeroed head per branch: [scan counter | finished blocks, flag]
static_assert(trx::kChainMaxBranchesHost * kBranchHead <= trx::kScratchZeroed, "zeroed head of the chain's arena");

int enqueue_chain(const trx_scenario_args* calls, const int* which, int n, double* const* out, hipStream_t st)
{
    const trx_scenario_args& s0 = calls[which[0]];
    const long N = s0.draw->N;
    if (N < 1 || N > 0x7fffffffL || n < 1 || n > trx::kChainMaxCalls) return TRX_ERR_ARG;
    trx::StreamLock turn(st);
    int nbr_total = 0;
    for (int i = 0; i < n; ++i) nbr_total += calls[which[i]].draw->planet ? 1 : 2;
    if (nbr_total > trx::kChainMaxBranchesHost) return TRX_ERR_ARG;
    const size_t branch_bytes = trx::chain_branch_scratch_bytes(N);

    Arena A;
    const size_t o_head = A.reserve(trx::kScratchZeroed);
    const size_t o_tab = A.reserve(sizeof(trx_draw_args) * (size_t)n);
    const size_t o_cnt = A.reserve(sizeof(int) * 2 * trx::kDrawMaxGroups * (size_t)n);
    struct CallOff { size_t cols, cols0, mask, mask2, prior, n, res, ws, pv, pi, idx[2], h[2]; };
    CallOff co[trx::kChainMaxCalls];
    for (int i = 0; i < n; ++i) {
        const trx_scenario_args& s = calls[which[i]];
        const int planet = s.draw->planet != 0, ncol = planet ? 11 : 14;
        co[i].cols = A.reserve(sizeof(double) * ncol * N);
        co[i].cols0 = A.reserve(sizeof(double) * 16);
        co[i].mask = A.reserve(N);
        co[i].mask2 = A.reserve(planet ? 0 : N);
        co[i].prior = A.reserve(s.want_prior ? sizeof(double) * N : 0);
        co[i].n = A.reserve(2 * sizeof(long));
        co[i].res = A.reserve(sizeof(double) * (2 * TRX_SCENARIO_OUT + 1));
        co[i].ws = A.reserve(sizeof(double) * 2 * 3 * kLmeParts);
        co[i].pv = A.reserve(sizeof(double) * 2 * kLmeParts);
        co[i].pi = A.reserve(sizeof(long) * 2 * 2 * kLmeParts);
        co[i].idx[0] = A.reserve(sizeof(int) * N);
        co[i].idx[1] = A.reserve(planet ? 0 : sizeof(int) * N);
        co[i].h[0] = A.reserve(sizeof(double) * N);
        co[i].h[1] = A.reserve(planet ? 0 : sizeof(double) * N);
    }
    const size_t o_branch = A.reserve(branch_bytes * (size_t)nbr_total);
    TRXS_HIP(trx::stream_scratch(st, 2, A.used, reinterpret_cast<void**>(&A.base)));

    // the calls' argument blocks, output pointers set, staged in pinned memory and copied to the device table
    trx_draw_args* stage = nullptr;
    void* stage_ticket = nullptr;
    TRXS_HIP(trx::pinned_stage_begin(st, sizeof(trx_draw_args) * (size_t)n, reinterpret_cast<void**>(&stage), &stage_ticket));
    trx::ChainFill fills[trx::kChainMaxCalls];
    trx::ChainBranch br[trx::kChainMaxBranchesHost];
    trx::ScenFinal fin[trx::kChainMaxBranchesHost];
    double* res_of[trx::kChainMaxCalls];
    bool copy_back[trx::kChainMaxCalls];
    int b_at = 0;
    for (int i = 0; i < n; ++i) {
        const trx_scenario_args& s = calls[which[i]];
        trx_draw_args d = *s.draw;
        const int planet = d.planet != 0, ncol = planet ? 11 : 14, nbr = planet ? 1 : 2;
        if (d.N != N || s.n_time != s0.n_time || s.time != s0.time || s.nsupersample != s0.nsupersample ||
            s.exptime != s0.exptime) return TRX_ERR_ARG;
        double* rec_dev = nullptr;
        {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, out[which[i]]) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
                rec_dev = static_cast<double*>(attr.devicePointer);
            else
                (void)hipGetLastError();
        }
        copy_back[i] = rec_dev == nullptr;
        double* res = rec_dev ? rec_dev : A.at<double>(co[i].res);
        res_of[i] = res;
        unsigned* state0 = reinterpret_cast<unsigned*>(A.base + o_head + (size_t)b_at * kBranchHead + 8);
        d.cols = A.at<double>(co[i].cols);
        d.mask = A.at<unsigned char>(co[i].mask);
        d.mask_twin = planet ? nullptr : A.at<unsigned char>(co[i].mask2);
        d.lnprior = s.want_prior ? A.at<double>(co[i].prior) : nullptr;
        d.flag = reinterpret_cast<int*>(state0 + 1);
        d.dump = nullptr;
        stage[i] = d;
        long* n_dev = A.at<long>(co[i].n);
        fills[i] = trx::ChainFill{A.at<int>(co[i].idx[0]), planet ? nullptr : A.at<int>(co[i].idx[1]), n_dev, A.at<double>(co[i].cols0)};
        for (int b = 0; b < nbr; ++b, ++b_at) {
            char* head = A.base + o_head + (size_t)b_at * kBranchHead;
            trx::ChainBranch& c = br[b_at];
            c.model = planet ? TRX_MODEL_TP : (b ? TRX_MODEL_EB_TWIN : TRX_MODEL_EB);
            c.flags = s.flags;
            c.twin = b;
            c.flux = s.flux;
            c.sigma = s.sigma;
            c.lnsigma = s.lnsigma;
            c.cols = d.cols;
            c.n_dev = n_dev + b;
            c.src_idx = A.at<int>(co[i].idx[b]);
            c.h = A.at<double>(co[i].h[b]);
            c.lnprior = d.lnprior;
            c.scratch = reinterpret_cast<double*>(A.base + o_branch + branch_bytes * (size_t)b_at);
            c.scan_count = reinterpret_cast<unsigned long long*>(head);
            c.ws = A.at<double>(co[i].ws) + (size_t)b * 3 * kLmeParts;
            c.amin_pv = A.at<double>(co[i].pv) + (size_t)b * kLmeParts;
            c.amin_pi = A.at<long>(co[i].pi) + (size_t)b * 2 * kLmeParts;
            trx::ScenFinal& f = fin[b_at];
            f = trx::ScenFinal{};
            f.idx = c.src_idx;
            f.cols = d.cols;
            f.cols0 = A.at<double>(co[i].cols0);
            f.cols0_stride = 1;
            f.dense = 1;
            f.N = N;
            f.n_total = N;
            f.ncol = ncol;
            f.branch = b;
            f.last_branch = (b == nbr - 1) ? 1 : 0;
            f.res = res + (size_t)b * TRX_SCENARIO_OUT;
            f.flag_out = (b == 0) ? res + 2 * TRX_SCENARIO_OUT : nullptr;
            f.state = reinterpret_cast<unsigned*>(head + 8);      // (branch 0: == state0, whose word 1 is the call's flag)
            c.fin = &f;
        }
    }
    trx_draw_args* dev_tab = A.at<trx_draw_args>(o_tab);
    {
        const hipError_t e = hipMemcpyAsync(dev_tab, stage, sizeof(trx_draw_args) * (size_t)n, hipMemcpyHostToDevice, st);
        trx::pinned_stage_end(st, stage_ticket);
        if (e != hipSuccess) return trx::fail_hip(e);
    }
    auto bail = [&](int rc) {
        (void)hipMemsetAsync(A.base + o_head, 0, trx::kScratchZeroed, st);
        return rc;
    };
    long per = 0;
    int groups = 0;
    if (int rc = trx::draw_chain(stage, dev_tab, n, A.at<int>(o_cnt), fills, &per, &groups, st)) return bail(rc);
    if (trx::knob_poison())          // tests: an unwritten row must show (include/trx_debug.h)
        for (int b = 0; b < nbr_total; ++b) TRXS_HIP(hipMemsetAsync(br[b].h, 0, sizeof(double) * (size_t)N, st));
    if (int rc = trx::lnl_lme_chain(br, nbr_total, s0.time, s0.n_time, N, s0.exptime, s0.nsupersample, st)) {
        // (kChainNotApplicable cannot happen in the production library -- trx_star_enqueue asked lnl_chain_applicable with
        // the same arguments and nothing it depends on can change; in the testing library a switch of trx_debug.h was
        // flipped by another thread in between: the draw kernels are already enqueued, so this is an error, not a fall-back)
        if (rc == trx::kChainNotApplicable) rc = trx::fail_hip(hipErrorInvalidValue);
        return bail(rc);
    }
    for (int i = 0; i < n; ++i)
        if (copy_back[i])
            TRXS_HIP(hipMemcpyAsync(out[which[i]], res_of[i], sizeof(double) * (2 * TRX_SCENARIO_OUT + 1), hipMemcpyDeviceToHost, st));
    return TRX_OK;
}

bool chain_enabled() { return trx::knob_star_chain() != 0; }

// may call j join a chain that starts with call i?  (same stream is the caller's business)
bool chain_compatible(const trx_scenario_args& a, const trx_scenario_args& b)
{
    return a.draw->N == b.draw->N && a.n_time == b.n_time && a.time == b.time && a.nsupersample == b.nsupersample &&
           a.exptime == b.exptime &&
           ((a.flags ^ b.flags) & trx::kChainSharedFlags) == 0;
}

}  // namespace

extern "C" int trx_scenario_enqueue(const trx_scenario_args* s, double* out, void* stream)
{
    if (!s || !s->draw || !out) return TRX_ERR_ARG;
    return enqueue(s, out, static_cast<hipStream_t>(stream));
}

extern "C" int trx_star_enqueue(const trx_scenario_args* calls, int n_calls, double* const* out, void* const* streams,
                                int* n_done)
{
    if (n_done) *n_done = 0;
    if (n_calls < 0 || (n_calls > 0 && (!calls || !out || !streams))) return TRX_ERR_ARG;
    for (int i = 0; i < n_calls; ++i)
        if (!calls[i].draw || !out[i]) return TRX_ERR_ARG;
    // Consecutive calls on one stream that share N, the time stamps, the exposure settings and the precision go into ONE
    // launch chain (enqueue_chain), up to kChainMaxCalls calls / kChainMaxBranchesHost branches at a time -- when the
    // bounded evaluation's passes apply to them and their scratch together stays below the budget (TRX_CHAIN_DRAWS,
    // default 2.5e7 draws' worth per chain: ~0.36 GB per 1e6 draws and call); everything else goes call by call.
    static const double draw_budget = trx::env_double("TRX_CHAIN_DRAWS", 2.5e7);
    int i = 0;
    while (i < n_calls) {
        hipStream_t st = static_cast<hipStream_t>(streams[i]);
        int which[trx::kChainMaxCalls];
        int n = 0, nbr = 0;
        if (chain_enabled() && calls[i].table_rows <= 1 &&
            trx::lnl_chain_applicable(calls[i].flags, calls[i].n_time, calls[i].draw->N, calls[i].nsupersample)) {
            for (int j = i; j < n_calls && n < trx::kChainMaxCalls; ++j) {
                if (streams[j] != streams[i] || calls[j].table_rows > 1 || !chain_compatible(calls[i], calls[j])) break;
                const int add = calls[j].draw->planet ? 1 : 2;
                if (nbr + add > trx::kChainMaxBranchesHost) break;
                if ((double)(n + 1) * (double)calls[i].draw->N > draw_budget && n > 0) break;
                which[n++] = j;
                nbr += add;
            }
        }
        if (n >= 2) {
            if (int rc = enqueue_chain(calls, which, n, out, st)) return rc;
            i += n;
        } else {
            if (int rc = enqueue(&calls[i], out[i], st)) return rc;
            i += 1;
        }
        if (n_done) *n_done = i;
    }
    return TRX_OK;
}

#ifdef TRX_TESTING
extern "C" int trx_set_star_chain(int on)
{
    trx::g_knob_star_chain = on ? 1 : 0;
    return TRX_OK;
}
#endif

extern "C" int trx_scenario_evidence(const trx_scenario_args* s, void* stream)
{
    if (!s || !s->draw || !s->out || !s->out_flag) return TRX_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    trx::StreamLock turn(st);              // the pinned staging record is the stream's, too
    double* pinned = nullptr;              // pinned, so that the copy is asynchronous
    TRXS_HIP(trx::stream_scratch(st, 3, sizeof(double) * (2 * TRX_SCENARIO_OUT + 1), reinterpret_cast<void**>(&pinned)));
    if (int rc = enqueue(s, pinned, st)) return rc;
    TRXS_HIP(hipStreamSynchronize(st));
    const int nbr = s->draw->planet ? 1 : 2;
    memcpy(s->out, pinned, (size_t)nbr * TRX_SCENARIO_OUT * sizeof(double));
    *s->out_flag = (int)pinned[2 * TRX_SCENARIO_OUT];
    return TRX_OK;
}

extern "C" size_t trx_scenario_args_size(void) { return sizeof(trx_scenario_args); }

#ifdef TRX_TESTING
extern "C" int trx_set_debug_poison(int on)
{
    trx::g_knob_poison = on ? 1 : 0;
    return TRX_OK;
}
#endif
