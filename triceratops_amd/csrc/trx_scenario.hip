// trx_scenario_evidence / trx_scenario_enqueue (include/trx.h): one lnZ_* call of calc_probs, end to end
// on the device, with no host synchronisation inside the call.
//
// Round 2's version strung the call together from 12 (planet) or 23 (binary) launches and synchronised
// the stream twice: the number of draws that pass the geometry mask sized the likelihood launch, so the
// host had to read it.  Here the count never leaves the device:
//   draw_kernel         the geometry mask(s) of every draw (trx_draw.hip: what does not feed a mask is not
//                       computed, no column is written); workgroup b takes draws [b per, (b + 1) per) and
//                       leaves its mask counts
//   compact_kernel      ordered compaction of the mask(s): a workgroup sums the counts of the workgroups
//                       before it and appends the indices of its own masked draws, ascending (the order
//                       numpy's / torch's nonzero gives) -> idx[branch][], n[branch]
//   fill_kernel         the parameter columns and the prior of the listed draws only (the 5-10 % that passed),
//                       recomputed from the same counter-based random numbers
//   rowc_kernel         } the likelihood of the masked draws, read IN PLACE from the draw kernel's columns
//   cells_kernel        } through idx (no gathered parameter block); the row count is read from n[branch]
//                         on the device and the grids are sized for a guess (trx_kernels.hip)
//   lme_partial_kernel  first pass of the log-mean-exp and of the search for the smallest chi^2, one pass
//   final_kernel        the evidence, the best draw (first of equals, NaN first: numpy's / torch's
//                       argmin), its columns, the masked count and the limb-darkening flag -> one record
// 7 launches for a planet scenario, 10 for a binary one (two branches), one 264-byte copy to the host,
// NO sync: a caller can enqueue every lnZ_* call of a calc_probs on a few streams and wait once
// (trx_scenario_enqueue); trx_scenario_evidence is the same followed by one hipStreamSynchronize.
// Every buffer lives in the stream's scratch (trx_internal.hpp).  Results are bit for bit those of the
// torch-operator chain of fused.py (same kernels on the same rows in the same order).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "../../include/trx.h"
#include "trx_device.hpp"
#include "trx_internal.hpp"

namespace {

using trx::Lme;

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

// Ordered compaction, second half.  grid = (groups, branches); workgroup (b, br) owns the draws
// [b per, (b + 1) per): their offset in the list is the sum of the counts of the workgroups before it
// (<= 4096 numbers, summed here: no scan kernel, no cross-workgroup hand-off), their order the draw index.
__global__ __launch_bounds__(256) void compact_kernel(const unsigned char* __restrict__ mask0,
                                                      const unsigned char* __restrict__ mask1, long N, long per,
                                                      const int* __restrict__ blk_cnt, int* __restrict__ idx0,
                                                      int* __restrict__ idx1, long* __restrict__ n_out)
{
    __shared__ long part[4];
    __shared__ int wave_off[4];
    const int br = blockIdx.y, groups = gridDim.x, b = blockIdx.x;
    const unsigned char* mask = br ? mask1 : mask0;
    int* idx = br ? idx1 : idx0;
    const int* cnt = blk_cnt + (long)br * groups;
    long before = 0;
    for (int j = threadIdx.x; j < b; j += 256) before += cnt[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = before;
    __syncthreads();
    long at = part[0] + part[1] + part[2] + part[3];
    if (b == groups - 1 && threadIdx.x == 0) n_out[br] = at + cnt[b];
    const long end = ((long)(b + 1) * per < N) ? (long)(b + 1) * per : N;
    for (long i0 = (long)b * per; i0 < end; i0 += 256) {
        const long i = i0 + threadIdx.x;
        const bool hit = i < end && mask[i] != 0;
        const unsigned long long m = __ballot(hit);
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        __syncthreads();                   // wave_off of the previous trip has been read
        if ((threadIdx.x & 63) == 0) wave_off[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        const int w = threadIdx.x >> 6;
        const int w0 = wave_off[0], w1 = wave_off[1], w2 = wave_off[2], w3 = wave_off[3];
        const int first = (w > 0 ? w0 : 0) + (w > 1 ? w1 : 0) + (w > 2 ? w2 : 0);
        if (hit) idx[at + first + below] = (int)i;
        at += w0 + w1 + w2 + w3;
    }
}

// One workgroup (one wavefront) per branch: the evidence from the log-mean-exp partials (the fold of
// lme_final_kernel, trx_kernels.hip: lane l takes partials l, l + 64, ... in order, then a fixed butterfly),
// the first minimum of chi^2 from the argmin partials, and the record
//   res[br][0 .. ncol)  the best draw's columns (draw 0 when no draw passed the mask)
//   res[br][ncol]       lnZ          res[br][ncol + 1]  the masked count
//   res[2][0]           the limb-darkening flag of the draw kernel (written by branch 0)
__global__ __launch_bounds__(64) void final_kernel(const double* __restrict__ ws, const double* __restrict__ amin_pv,
                                                   const long* __restrict__ amin_pi, const long* __restrict__ n_dev,
                                                   const int* __restrict__ idx0, const int* __restrict__ idx1,
                                                   const double* __restrict__ cols, long N, int ncol, long n_total,
                                                   const int* __restrict__ flag, double* __restrict__ res)
{
    const int br = blockIdx.x, lane = threadIdx.x;
    const long n = n_dev[br];
    const int nparts = trx::lme_blocks(n);
    const double* w = ws + (size_t)br * 3 * kLmeParts;
    const double* pv = amin_pv + (size_t)br * kLmeParts;
    const long* pi = amin_pi + (size_t)br * kLmeParts;
    const int* idx = br ? idx1 : idx0;
    Lme t{-INFINITY, 0.0, 0};
    double bv = INFINITY;
    long bi = -1;
    if (n > 0) {
        for (int i = lane; i < nparts; i += 64) {
            Lme o{w[3 * i], w[3 * i + 1], w[3 * i + 2] != 0.0};
            trx::lme_merge(t, o);
            const long oi = pi[i];
            if (oi >= 0 && (bi < 0 || trx::argmin_before(pv[i], oi, bv, bi))) { bv = pv[i]; bi = oi; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Lme other;
        other.m = __shfl_xor(t.m, o, 64);
        other.s = __shfl_xor(t.s, o, 64);
        other.pinf = __shfl_xor(t.pinf, o, 64);
        trx::lme_merge(t, other);
        const double ov = __shfl_xor(bv, o, 64);
        const long oi = __shfl_xor(bi, o, 64);
        if (oi >= 0 && (bi < 0 || trx::argmin_before(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    double lnz;
    if (t.pinf) lnz = INFINITY;                                   // _numerics.py:46-47
    else if (t.m == -INFINITY) lnz = -INFINITY;                   // :49-50
    else lnz = log(t.s) + t.m - log((double)n_total);             // :51
    const long best = (bi >= 0) ? (long)idx[bi] : 0;
    double* r = res + (size_t)br * TRX_SCENARIO_OUT;
    if (lane < ncol) r[lane] = cols[(long)lane * N + best];
    if (lane == ncol) r[ncol] = lnz;
    if (lane == ncol + 1) r[ncol + 1] = (double)n;
    if (br == 0 && lane == 63) res[2 * TRX_SCENARIO_OUT] = (double)flag[0];
}

int enqueue(const trx_scenario_args* s, double* out_host, hipStream_t st)
{
    trx_draw_args d = *s->draw;
    const long N = d.N;
    if (N < 1 || N > 0x7fffffffL) return TRX_ERR_ARG;
    const int planet = d.planet != 0;
    const int ncol = planet ? 11 : 14, nbr = planet ? 1 : 2;
    trx::StreamLock turn(st);              // the whole call is enqueued back to back on the stream's scratch

    Arena A;
    const size_t o_cols = A.reserve(sizeof(double) * ncol * N), o_mask = A.reserve(N), o_mask2 = A.reserve(planet ? 0 : N),
                 o_prior = A.reserve(s->want_prior ? sizeof(double) * N : 0), o_flag = A.reserve(sizeof(int)),
                 o_n = A.reserve(2 * sizeof(long)), o_res = A.reserve(sizeof(double) * (2 * TRX_SCENARIO_OUT + 1)),
                 o_ws = A.reserve(sizeof(double) * 2 * 3 * kLmeParts), o_pv = A.reserve(sizeof(double) * 2 * kLmeParts),
                 o_pi = A.reserve(sizeof(long) * 2 * kLmeParts), o_cnt = A.reserve(sizeof(int) * 2 * trx::kDrawMaxGroups),
                 o_idx0 = A.reserve(sizeof(int) * N), o_idx1 = A.reserve(planet ? 0 : sizeof(int) * N),
                 o_h0 = A.reserve(sizeof(double) * N), o_h1 = A.reserve(planet ? 0 : sizeof(double) * N);
    TRXS_HIP(trx::stream_scratch(st, 1, A.used, reinterpret_cast<void**>(&A.base)));
    d.cols = A.at<double>(o_cols);
    d.mask = A.at<unsigned char>(o_mask);
    d.mask_twin = planet ? nullptr : A.at<unsigned char>(o_mask2);
    d.lnprior = s->want_prior ? A.at<double>(o_prior) : nullptr;
    d.flag = A.at<int>(o_flag);
    d.dump = nullptr;
    long* n_dev = A.at<long>(o_n);
    double* res = A.at<double>(o_res);
    int* idx[2] = {A.at<int>(o_idx0), planet ? nullptr : A.at<int>(o_idx1)};
    double* h[2] = {A.at<double>(o_h0), planet ? nullptr : A.at<double>(o_h1)};
    TRXS_HIP(hipMemsetAsync(d.flag, 0, sizeof(int), st));
    long per = 0;
    int groups = 0;
    if (int rc = trx::draw_counted(d, A.at<int>(o_cnt), &per, &groups, st)) return rc;
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)groups, (unsigned)nbr), dim3(256), 0, st, d.mask, d.mask_twin, N, per,
                       A.at<int>(o_cnt), idx[0], idx[1], n_dev);
    if (int rc = trx::fill_draws(d, idx[0], idx[1], n_dev, st)) return rc;
    for (int b = 0; b < nbr; ++b) {
        const int model = planet ? TRX_MODEL_TP : (b ? TRX_MODEL_EB_TWIN : TRX_MODEL_EB);
        const double* bounds = nullptr;
        if (int rc = trx::lnl_draws(model, s->flags, s->time, s->flux, s->n_time, s->sigma, d.cols, N, n_dev + b, idx[b], N,
                                    b, s->exptime, s->nsupersample, h[b], d.lnprior, s->lnsigma, &bounds, st))
            return rc;
        if (int rc = trx::lme_draws(h[b], d.lnprior, s->lnsigma, N, n_dev + b, idx[b],
                                    A.at<double>(o_ws) + (size_t)b * 3 * kLmeParts, A.at<double>(o_pv) + (size_t)b * kLmeParts,
                                    A.at<long>(o_pi) + (size_t)b * kLmeParts, bounds, st))
            return rc;
    }
    hipLaunchKernelGGL(final_kernel, dim3((unsigned)nbr), dim3(64), 0, st, A.at<double>(o_ws), A.at<double>(o_pv),
                       A.at<long>(o_pi), n_dev, idx[0], idx[1], d.cols, N, ncol, N, d.flag, res);
    TRXS_HIP(hipGetLastError());
    TRXS_HIP(hipMemcpyAsync(out_host, res, sizeof(double) * (2 * TRX_SCENARIO_OUT + 1), hipMemcpyDeviceToHost, st));
    return TRX_OK;
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
    for (int i = 0; i < n_calls; ++i) {
        if (!calls[i].draw || !out[i]) return TRX_ERR_ARG;
        if (int rc = enqueue(&calls[i], out[i], static_cast<hipStream_t>(streams[i]))) return rc;
        if (n_done) *n_done = i + 1;
    }
    return TRX_OK;
}

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
