// trx_scenario_evidence (include/trx.h): one lnZ_* call of calc_probs, end to end on the device.
//
// The Python host of round 2 strung a lnZ_* call together from ~30 launches (draw kernel, torch
// nonzero / index_select / argmin / cat, likelihood, log-mean-exp) and two host syncs; with the
// kernels where they are now that glue -- 25 torch operators and their Python dispatch under the
// GIL -- is what bounds calc_probs_many on many host threads.  Here the whole call is one C entry
// point: draw kernel -> ordered compaction of the geometry mask(s) (rocPRIM select) -> gather of
// the masked parameter block -> rowc_kernel + cells_kernel -> log-mean-exp -> first-minimum of
// chi^2 -> the best draw's columns, lnZ and the masked count in ONE small device-to-host copy.
// Two stream syncs per call (the masked counts size the likelihood launches; the result), no
// Python in between: ctypes releases the GIL for the duration of the call.  Every buffer lives in
// the stream's scratch (trx_internal.hpp): hipMallocAsync / hipFreeAsync cost ~60 us per call on
// this stack (profiles/r02_g_native_call.txt), 14 buffers a call more than the kernels saved.
//
// It returns what calc_probs keeps of a scenario (triceratops.py:804-817 ...: the best draw and
// lnZ); the 100-row best-fit table of a direct lnZ_* call stays on the torch path of fused.py.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>

#include <string.h>

#include "../../include/trx.h"
#include "trx_internal.hpp"

namespace {

#define TRXS_HIP(call)                                   \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return TRX_ERR_HIP;        \
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

// block[p][i] = cols[p][idx[i]] for the nblk likelihood parameters; the twin branch runs at twice
// the period with the semi-major axis of that period (marginal_likelihoods.py:300-339: row 2 = P,
// row 4 = a, a at 2 P_orb is column 11 of the draw kernel's block)
__global__ __launch_bounds__(256) void gather_block_kernel(const double* __restrict__ cols, long N,
                                                           const long* __restrict__ idx, const long* __restrict__ count,
                                                           int nblk, int twin, const double* __restrict__ lnprior,
                                                           double* __restrict__ block, double* __restrict__ lp)
{
    const long n = *count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long src = idx[i];
        for (int p = 0; p < nblk; ++p) {
            double v = cols[(long)p * N + src];
            if (twin && p == 2) v *= 2.0;
            if (twin && p == 4) v = cols[11L * N + src];
            block[(long)p * n + i] = v;
        }
        if (lp) lp[i] = lnprior[src];
    }
}

// torch.argmin's order: NaN before everything, then the smallest value; ties -> the lowest index
__device__ __forceinline__ bool before(double a, long ia, double b, long ib)
{
    const bool na = a != a, nb = b != b;
    if (na != nb) return na;
    if (!na && a != b) return a < b;
    return ia < ib;
}

// The masked draw with the smallest chi^2, two stages: kArgminBlocks partial (value, position) pairs,
// then one block over the partials; best[0] = its index into the N draws, draw 0 when no draw
// passed the mask (the table row the torch path fills in then).
constexpr int kArgminBlocks = 128;

__device__ __forceinline__ void argmin_block(double v, long at, double* sv, long* si)
{
    sv[threadIdx.x] = v;
    si[threadIdx.x] = at;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const long j = si[threadIdx.x + o];
            if (j >= 0 && (si[threadIdx.x] < 0 || before(sv[threadIdx.x + o], j, sv[threadIdx.x], si[threadIdx.x]))) {
                sv[threadIdx.x] = sv[threadIdx.x + o];
                si[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void argmin_partial_kernel(const double* __restrict__ h, const long* __restrict__ count,
                                                             double* __restrict__ pv, long* __restrict__ pi)
{
    __shared__ double sv[256];
    __shared__ long si[256];
    const long n = *count;
    double v = INFINITY;
    long at = -1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double x = h[i];
        if (at < 0 || before(x, i, v, at)) { v = x; at = i; }
    }
    argmin_block(v, at, sv, si);
    if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}

__global__ __launch_bounds__(kArgminBlocks) void argmin_final_kernel(const double* __restrict__ pv, const long* __restrict__ pi,
                                                                     const long* __restrict__ idx, long* __restrict__ best)
{
    __shared__ double sv[kArgminBlocks];
    __shared__ long si[kArgminBlocks];
    argmin_block(pv[threadIdx.x], pi[threadIdx.x], sv, si);
    if (threadIdx.x == 0) best[0] = (si[0] >= 0) ? idx[si[0]] : 0;
}

// res = [the best draw's ncol columns, lnZ, masked count]
__global__ void collect_kernel(const double* __restrict__ cols, long N, int ncol, const long* __restrict__ best,
                               const double* __restrict__ lnz, const long* __restrict__ count, double* __restrict__ res)
{
    const int c = threadIdx.x;
    if (c < ncol) res[c] = cols[(long)c * N + best[0]];
    if (c == ncol) res[ncol] = lnz[0];
    if (c == ncol + 1) res[ncol + 1] = (double)count[0];
}

}  // namespace

extern "C" int trx_scenario_evidence(const trx_scenario_args* s, void* stream)
{
    if (!s || !s->draw || !s->out || !s->out_flag) return TRX_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    trx_draw_args d = *s->draw;
    const long N = d.N;
    if (N < 1 || N > 0x7fffffffL) return TRX_ERR_ARG;
    const int planet = d.planet != 0;
    const int ncol = planet ? 11 : 14, nblk = planet ? 10 : 11, nbr = planet ? 1 : 2;
    const size_t ws_bytes = trx_workspace_bytes();
    hipcub::CountingInputIterator<long> iota(0);
    size_t tmp_bytes = 0;
    TRXS_HIP(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, iota, (unsigned char*)nullptr, (long*)nullptr,
                                           (long*)nullptr, (int)N, st));

    // buffers sized by N, in the stream's scratch (no allocator traffic once it has grown)
    Arena A;
    const size_t o_cols = A.reserve(sizeof(double) * ncol * N), o_mask = A.reserve(N), o_mask2 = A.reserve(planet ? 0 : N),
                 o_prior = A.reserve(s->want_prior ? sizeof(double) * N : 0), o_flag = A.reserve(sizeof(int)),
                 o_cnt = A.reserve(4 * sizeof(long)), o_res = A.reserve(sizeof(double) * (2 * TRX_SCENARIO_OUT + 2)),
                 o_ws = A.reserve(ws_bytes), o_tmp = A.reserve(tmp_bytes), o_idx0 = A.reserve(sizeof(long) * N),
                 o_idx1 = A.reserve(planet ? 0 : sizeof(long) * N),
                 o_pv = A.reserve(sizeof(double) * kArgminBlocks), o_pi = A.reserve(sizeof(long) * kArgminBlocks);
    TRXS_HIP(trx::stream_scratch(st, 1, A.used, reinterpret_cast<void**>(&A.base)));
    d.cols = A.at<double>(o_cols);
    d.mask = A.at<unsigned char>(o_mask);
    d.mask_twin = planet ? nullptr : A.at<unsigned char>(o_mask2);
    d.lnprior = s->want_prior ? A.at<double>(o_prior) : nullptr;
    d.flag = A.at<int>(o_flag);
    d.dump = nullptr;
    long* cnt = A.at<long>(o_cnt);             // [2] masked counts, [2] best indices
    double* res = A.at<double>(o_res);         // [2][TRX_SCENARIO_OUT] + lnz[2]
    long* idx[2] = {A.at<long>(o_idx0), planet ? nullptr : A.at<long>(o_idx1)};
    TRXS_HIP(hipMemsetAsync(d.flag, 0, sizeof(int), st));
    if (int rc = trx_draw_scenario(&d, st)) return rc;

    // ordered compaction of the mask(s): the indices of the masked draws, ascending
    for (int b = 0; b < nbr; ++b)
        TRXS_HIP(hipcub::DeviceSelect::Flagged(A.at<char>(o_tmp), tmp_bytes, iota, b ? d.mask_twin : d.mask, idx[b],
                                               cnt + b, (int)N, st));
    char* pinned = nullptr;                    // results on their way back: pinned, so the copies are asynchronous
    TRXS_HIP(trx::stream_scratch(st, 3, 512, reinterpret_cast<void**>(&pinned)));
    long* n_host = reinterpret_cast<long*>(pinned);
    double* out_host = reinterpret_cast<double*>(pinned + 64);
    int* flag_host = reinterpret_cast<int*>(pinned + 64 + sizeof(double) * 2 * TRX_SCENARIO_OUT);
    TRXS_HIP(hipMemcpyAsync(n_host, cnt, 2 * sizeof(long), hipMemcpyDeviceToHost, st));
    TRXS_HIP(hipStreamSynchronize(st));
    const long n_br[2] = {n_host[0], planet ? 0 : n_host[1]};

    // buffers sized by the masked counts
    Arena B;
    size_t o_block[2], o_h[2], o_lp[2];
    for (int b = 0; b < nbr; ++b) {
        o_block[b] = B.reserve(sizeof(double) * nblk * n_br[b]);
        o_h[b] = B.reserve(sizeof(double) * (n_br[b] > 0 ? n_br[b] : 1));
        o_lp[b] = B.reserve(s->want_prior ? sizeof(double) * n_br[b] : 0);
    }
    TRXS_HIP(trx::stream_scratch(st, 2, B.used, reinterpret_cast<void**>(&B.base)));
    for (int b = 0; b < nbr; ++b) {
        const long n = n_br[b];
        const int model = planet ? TRX_MODEL_TP : (b ? TRX_MODEL_EB_TWIN : TRX_MODEL_EB);
        double* block = B.at<double>(o_block[b]);
        double* h = B.at<double>(o_h[b]);
        double* lp = s->want_prior ? B.at<double>(o_lp[b]) : nullptr;
        if (n > 0) {
            const unsigned grid = (unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
            hipLaunchKernelGGL(gather_block_kernel, dim3(grid), dim3(256), 0, st, d.cols, N, idx[b], cnt + b,
                               nblk, b, d.lnprior, block, lp);
        }
        double* lnz = res + 2 * TRX_SCENARIO_OUT + b;
        if (int rc = trx_lnz_scenario(model, s->flags, s->time, s->flux, s->n_time, s->sigma, block, n,
                                      s->exptime, s->nsupersample, lp, N, s->lnsigma, h, lnz, A.at<char>(o_ws),
                                      ws_bytes, st))
            return rc;
        hipLaunchKernelGGL(argmin_partial_kernel, dim3(kArgminBlocks), dim3(256), 0, st, h, cnt + b,
                           A.at<double>(o_pv), A.at<long>(o_pi));
        hipLaunchKernelGGL(argmin_final_kernel, dim3(1), dim3(kArgminBlocks), 0, st, A.at<double>(o_pv),
                           A.at<long>(o_pi), idx[b], cnt + 2 + b);
        hipLaunchKernelGGL(collect_kernel, dim3(1), dim3(64), 0, st, d.cols, N, ncol, cnt + 2 + b, lnz, cnt + b,
                           res + b * TRX_SCENARIO_OUT);
    }
    TRXS_HIP(hipGetLastError());
    TRXS_HIP(hipMemcpyAsync(out_host, res, (size_t)nbr * TRX_SCENARIO_OUT * sizeof(double), hipMemcpyDeviceToHost, st));
    TRXS_HIP(hipMemcpyAsync(flag_host, d.flag, sizeof(int), hipMemcpyDeviceToHost, st));
    TRXS_HIP(hipStreamSynchronize(st));
    memcpy(s->out, out_host, (size_t)nbr * TRX_SCENARIO_OUT * sizeof(double));
    *s->out_flag = *flag_host;
    return TRX_OK;
}

extern "C" size_t trx_scenario_args_size(void) { return sizeof(trx_scenario_args); }
