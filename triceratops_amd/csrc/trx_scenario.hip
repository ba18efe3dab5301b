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
//                        random numbers
//   rowc_kernel          } the likelihood of the masked draws, read IN PLACE from the draw kernel's columns through idx
//   (sec_scan_kernel)    } (no gathered parameter block); the row count is read from n[branch] on the device and the
//   cells_kernel         } grids are sized for a guess (trx_kernels.hip)
//   lme_partial_kernel   first pass of the log-mean-exp and of the search for the smallest chi^2 in one pass; the block
//                        that finishes LAST folds the partials into the record: the evidence, the best draw (first of
//                        equals, NaN first: numpy's / torch's argmin), its columns, the masked count and the
//                        limb-darkening flag (scenario_final, trx_device.hpp), written straight into the caller's
//                        pinned record
// (with the bounded evaluation cells_kernel is up to five launches: pilot rows, pilot_stats_kernel, depth_screen_kernel,
// probe pass, the rows left alive -- trx_kernels.hip)
// 5 launches for a planet scenario (7 in round 3), 9 for a binary one (two branches; 10), no memset, no copy, NO
// sync: a caller can enqueue every lnZ_* call of a calc_probs on a few streams and wait once (trx_scenario_enqueue);
// trx_scenario_evidence is the same followed by one hipStreamSynchronize.  Every buffer lives in the stream's scratch
// (trx_internal.hpp), behind a small persistent block (the finished-block counter of the last stage, the draw
// kernel's flag) that the kernels leave at zero.  Results are bit for bit those of the torch-operator chain of
// fused.py (same arithmetic on the same rows in the same order).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <atomic>

#include "../../include/trx.h"
#include "trx_device.hpp"
#include "trx_internal.hpp"

namespace {

std::atomic<int> g_poison{0};         // trx_set_debug_poison (tests)

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

int enqueue(const trx_scenario_args* s, double* out_host, hipStream_t st)
{
    trx_draw_args d = *s->draw;
    const long N = d.N;
    if (N < 1 || N > 0x7fffffffL) return TRX_ERR_ARG;
    const int planet = d.planet != 0;
    const int ncol = planet ? 11 : 14, nbr = planet ? 1 : 2;
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
                 o_n = A.reserve(2 * sizeof(long)), o_res = A.reserve(sizeof(double) * (2 * TRX_SCENARIO_OUT + 1)),
                 o_ws = A.reserve(sizeof(double) * 2 * 3 * kLmeParts), o_pv = A.reserve(sizeof(double) * 2 * kLmeParts),
                 o_pi = A.reserve(sizeof(long) * 2 * kLmeParts), o_cnt = A.reserve(sizeof(int) * 2 * trx::kDrawMaxGroups),
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
    if (int rc = trx::compact_fill(d, per, groups, A.at<int>(o_cnt), idx[0], idx[1], n_dev, st)) return bail(rc);
    if (g_poison.load(std::memory_order_relaxed))          // tests: an unwritten row must show (include/trx.h)
        for (int b = 0; b < nbr; ++b) TRXS_HIP(hipMemsetAsync(h[b], 0, sizeof(double) * (size_t)N, st));
    for (int b = 0; b < nbr; ++b) {
        const int model = planet ? TRX_MODEL_TP : (b ? TRX_MODEL_EB_TWIN : TRX_MODEL_EB);
        const double* bounds = nullptr;
        if (int rc = trx::lnl_draws(model, s->flags, s->time, s->flux, s->n_time, s->sigma, d.cols, N, n_dev + b, idx[b], N,
                                    b, s->exptime, s->nsupersample, h[b], d.lnprior, s->lnsigma, &bounds, st))
            return bail(rc);
        trx::ScenFinal fin{};
        fin.idx = idx[b];
        fin.cols = d.cols;
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
                                    A.at<long>(o_pi) + (size_t)b * kLmeParts, bounds, fin, st))
            return bail(rc);
    }
    if (!rec_dev)
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

extern "C" int trx_set_debug_poison(int on)
{
    g_poison = on ? 1 : 0;
    return TRX_OK;
}
