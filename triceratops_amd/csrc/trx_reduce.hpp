// The HBM-bound reductions: chi2_grid_kernel (row reduction over a materialised grid) and the log-mean-exp kernels
// (lme_partial_kernel / lme_partial_kernel_star with the scenario's final stage, lme_final_kernel).  Included by
// trx_kernels.hip only.
#pragma once
#include "trx_cells.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// 0.5 * sum_t (flux_t - model[r][t])^2 / sigma^2, one wavefront per row, 16 B/lane loads.
__global__ __launch_bounds__(256) void chi2_grid_kernel(const double* __restrict__ flux,
                                                        const double* __restrict__ grid,
                                                        int n_time, long n, double sigma,
                                                        double* __restrict__ out, int vec_ok)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const double s2 = sigma * sigma;
    for (long r = wave0; r < n; r += nwaves) {
        const double* row = grid + (size_t)r * n_time;
        double acc = 0.0;
        if (vec_ok) {
            typedef double dvec2 __attribute__((ext_vector_type(2)));
            const dvec2* row2 = reinterpret_cast<const dvec2*>(row);
            const dvec2* fl2 = reinterpret_cast<const dvec2*>(flux);
            const int nv = n_time >> 1;
            for (int j = lane; j < nv; j += 64) {
                const dvec2 m = __builtin_nontemporal_load(&row2[j]);
                const dvec2 f = fl2[j];
                const double d0 = f.x - m.x, d1 = f.y - m.y;
                acc += (d0 * d0) / s2;
                acc += (d1 * d1) / s2;
            }
        } else {
            for (int j = lane; j < n_time; j += 64) {
                const double d = flux[j] - row[j];
                acc += (d * d) / s2;
            }
        }
        const double tot = wave_sum(acc);
        if (lane == 0) out[r] = 0.5 * tot;
    }
}

// ---------------------------------------------------------------------------------------
// log-mean-exp.  Partial state per thread: running max m (finite or -inf), s = sum exp(x - m),
// pinf = saw +inf.  NaN and -inf carry zero weight (_numerics.py:48).
// (struct Lme, lme_merge: trx_device.hpp)

// x_i = c0 - h_i + lnprior_i (fused lnZ tail) when h != null, else x_i = logw_i
__device__ __forceinline__ double lme_value(const double* logw, const double* h,
                                            const double* lnprior, double c0, long i)
{
    if (!h) return logw[i];  // plain log-weights
    double x = c0 - h[i];
    if (lnprior) x += lnprior[i];
    return x;
}

// fold four values into the running (max, sum) state: one rescale, exps only for terms that can
// reach the sum: s >= 1 always (the max contributes exp(0)), so a term with d = x - max < -80
// is < 1.8e-35 and even 2^60 of them stay below fp64 resolution of s (also covers -inf)
__device__ __forceinline__ void lme_fold4(Lme& st, double x0, double x1, double x2, double x3)
{
    double x[4] = {x0, x1, x2, x3};
    double cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double v = x[u];
        if (v == INFINITY) { st.pinf = 1; v = -INFINITY; }
        if (!(v == v)) v = -INFINITY;
        x[u] = v;
        cm = fmax(cm, v);
    }
    if (cm == -INFINITY) return;
    if (cm > st.m) {
        const double d = st.m - cm;
        st.s = (d > -80.0) ? st.s * exp(d) : 0.0;
        st.m = cm;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double d = x[u] - st.m;
        if (d > -80.0) st.s += exp(d);
    }
}

// Loads per lane per trip (16 B each) and whether the next trip's loads are issued before the
// current trip is folded (profiles/r02/lme_variants.txt)
#ifndef TRX_LME_LOADS
#define TRX_LME_LOADS 2
#endif
#ifndef TRX_LME_PREFETCH
#define TRX_LME_PREFETCH 0
#endif
// slots of the per-thread queue of terms that can still count
#ifndef TRX_LME_QUEUE
#define TRX_LME_QUEUE 8
#endif


// SCEN (trx_scenario_evidence): the element count comes from the device (n_dev; the grid was sized for
// its upper bound and the blocks beyond lme_blocks(n) leave at once), lnprior is indexed by the draw
// (src_idx: the masked draws are not gathered), and the pass also finds the first minimum of h
// (argmin partials behind the 3 * 2048 sums of the workspace).
constexpr int kLmeMaxBlocks = 2048;
// same value for the search of the smallest chi^2 (NaN equals NaN there: torch.argmin's order)
__device__ __forceinline__ bool argmin_same(double a, double b) { return a == b || (a != a && b != b); }

// merge of two (value, first position, number of rows holding that value) states of the search
__device__ __forceinline__ void argmin_merge(double& v, long& i, long& c, double ov, long oi, long oc)
{
    if (oi < 0) return;
    if (i < 0) { v = ov; i = oi; c = oc; return; }
    if (argmin_same(v, ov)) { c += oc; i = oi < i ? oi : i; return; }
    if (argmin_before(ov, oi, v, i)) { v = ov; i = oi; c = oc; }
}

template <bool SCEN>
__device__ __forceinline__ void lme_partial_body(const double* __restrict__ logw,
                                                 const double* __restrict__ h,
                                                 const double* __restrict__ lnprior,
                                                 double c0, long n, int vec_ok,
                                                 double* __restrict__ ws,
                                                 const long* __restrict__ n_dev,
                                                 const int* __restrict__ src_idx,
                                                 double* __restrict__ amin_pv, long* __restrict__ amin_pi,
                                                 const double* __restrict__ bounds_base, const ScenFinal& fin)
{
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    Lme st{-INFINITY, 0.0, 0};
    unsigned nblocks = gridDim.x;
    // SCEN after a bounded evaluation (cells_kernel<PRUNE>): the launch header holds the largest log-weight M
    // of the call, and every term below M - 90 is taken as -inf.  Such a term carries no weight either way;
    // but WHICH rows were abandoned (and report a bound instead of their value) depends on timing, and a term
    // that is large against a thread's running maximum steers the fold below (queues, the census): filtered,
    // the fold sees the same numbers every run -- the rows within 90 of M are never abandoned.
    double floor_x = -INFINITY;
    if (SCEN) {
        n = *n_dev;
        nblocks = (unsigned)lme_blocks(n);
        if (blockIdx.x >= nblocks) return;
        if (bounds_base) floor_x = bounds_base[n * kRowDoubles + kHdrXmax] - 90.0;
    }
    double amin_v = INFINITY;          // SCEN: this thread's first minimum of h ...
    long amin_i = -1, amin_c = 0;      // ... and the number of its rows that hold that value
    bool unwritten = false;            // SCEN: a row still carries rowc_kernel's "never written" mark
    long stride = (long)nblocks * blockDim.x;
    long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec_ok) {
        // 16 B per lane per load, TRX_LME_LOADS independent loads per trip (issuing the next
        // trip's loads before the fold, TRX_LME_PREFETCH, measured slower: profiles/r02/lme_variants.txt).
        // The fold itself leans on IEEE max: fmax ignores a NaN operand, so NaN never reaches the
        // running maximum, `x - max > -80` is false for NaN and for -inf, and a +inf drives the
        // maximum to +inf (detected once, after the loop) -- no per-value inf / NaN tests.
        // Terms within 80 of the running maximum are rare once the maximum has settled (a few
        // per cent of a broad log-likelihood distribution) but with 64 lanes x 4-8 values per trip
        // some lane nearly always holds one, and the whole wave would run exp several times per
        // trip.  So a term that can still count is parked in a per-thread queue in LDS and the
        // exps run over the queues only when one of them fills.  The parked values are raw, so a
        // later, larger maximum needs no bookkeeping.
        constexpr int kL = TRX_LME_LOADS, kV = 2 * kL;
        // Every block streams ONE contiguous segment of the vector (a multiple of 4 KB), its waves 1 KB
        // apart: 6.35 TB/s on the 3.2 GB stress vector against 5.75 with the usual grid-stride loop,
        // where a block's consecutive loads are gridDim x 4 KB apart (profiles/r02/lme_variants.txt).
        const long nv_all = n >> 1;
        const long seg = ((nv_all + nblocks - 1) / nblocks + 255) & ~255L;
        const long seg0 = (long)blockIdx.x * seg;
        const long nv = (seg0 + seg < nv_all) ? (seg0 + seg) : nv_all;
        stride = blockDim.x;
        tid = seg0 + threadIdx.x;
        const dvec2* src = reinterpret_cast<const dvec2*>(h ? h : logw);
        const dvec2* pri = reinterpret_cast<const dvec2*>(lnprior);
        constexpr int kQ = (kV > 4 && TRX_LME_QUEUE < 12) ? 12 : TRX_LME_QUEUE;
        __shared__ double qbuf[kQ][256];
        int qc = 0;
        auto flush = [&]() {
            for (int i = 0; i < kQ; ++i) {
                if (i < qc) {
                    const double d = qbuf[i][threadIdx.x] - st.m;
                    if (d > -80.0) st.s += exp(d);
                }
            }
            qc = 0;
        };
        // unconditional loads at clamped indices (a predicated load forces vmcnt(0) at the join);
        // slots past the end are masked to -inf when the trip is folded
        auto fetch = [&](long v0, dvec2* a, dvec2* p) {
#pragma unroll
            for (int u = 0; u < kL; ++u) {
                long j = v0 + u * stride;
                j = (j < nv) ? j : (nv - 1);
                a[u] = __builtin_nontemporal_load(&src[j]);
                if (SCEN) {
                    if (lnprior) {
                        // the prior of rows 2 j, 2 j + 1: dense (one 16-byte read; the twin branch's run is stored from
                        // the top down), or by draw index through the list
                        if (fin.dense) {
                            if (fin.branch) { p[u].x = lnprior[fin.N - 1 - 2 * j]; p[u].y = lnprior[fin.N - 2 - 2 * j]; }
                            else { p[u].x = lnprior[2 * j]; p[u].y = lnprior[2 * j + 1]; }
                        } else {
                            p[u].x = lnprior[src_idx[2 * j]];
                            p[u].y = lnprior[src_idx[2 * j + 1]];
                        }
                    }
                } else if (pri) p[u] = __builtin_nontemporal_load(&pri[j]);
            }
        };
        dvec2 cur[kL], curp[kL], nxt[kL], nxtp[kL];
        if (tid < nv) fetch(tid, cur, curp);
        for (long v = tid; v < nv; v += kL * stride) {
            const long vn = v + kL * stride;
            if (TRX_LME_PREFETCH && vn < nv) fetch(vn, nxt, nxtp);
            double x[kV];
#pragma unroll
            for (int u = 0; u < kL; ++u) {
                dvec2 a = cur[u];
                const bool ok = v + u * stride < nv;
                if (SCEN && ok) {
                    // (a thread meets its elements in ascending order: a later one wins only when strictly before)
                    const long e = 2 * (v + u * stride);
                    if (amin_i < 0 || argmin_before(a.x, e, amin_v, amin_i)) { amin_v = a.x; amin_i = e; amin_c = 1; }
                    else if (argmin_same(a.x, amin_v)) ++amin_c;
                    if (argmin_before(a.y, e + 1, amin_v, amin_i)) { amin_v = a.y; amin_i = e + 1; amin_c = 1; }
                    else if (argmin_same(a.y, amin_v)) ++amin_c;
                    unwritten = unwritten || (unsigned long long)__double_as_longlong(a.x) == kUnwrittenBits ||
                                (unsigned long long)__double_as_longlong(a.y) == kUnwrittenBits;
                }
                if (h) {
                    a = c0 - a;
                    if (pri) a += curp[u];
                }
                if (SCEN) {
                    a.x = (a.x < floor_x) ? -INFINITY : a.x;
                    a.y = (a.y < floor_x) ? -INFINITY : a.y;
                }
                x[2 * u] = ok ? a.x : -INFINITY;
                x[2 * u + 1] = ok ? a.y : -INFINITY;
            }
            double cm = x[0];
#pragma unroll
            for (int u = 1; u < kV; ++u) cm = fmax(cm, x[u]);
            if (cm > st.m) {
                const double d = st.m - cm;
                st.s = (d > -80.0) ? st.s * exp(d) : 0.0;
                st.m = cm;
            }
            bool live[kV];
            int crowd = 0;
#pragma unroll
            for (int u = 0; u < kV; ++u) {
                live[u] = x[u] - st.m > -80.0;
#ifndef TRX_LME_NOCROWD
                crowd += __popcll(__ballot(live[u]));
#endif
            }
            if (crowd > 24 * kV) {
                // a narrow distribution: most terms count, nothing to gain from parking them
#pragma unroll
                for (int u = 0; u < kV; ++u)
                    if (live[u]) st.s += exp(x[u] - st.m);
            } else {
                if (__any(qc > kQ - kV)) flush();
#pragma unroll
                for (int u = 0; u < kV; ++u) {
                    if (live[u]) {
                        qbuf[qc][threadIdx.x] = x[u];
                        ++qc;
                    }
                }
            }
            if (TRX_LME_PREFETCH) {
#pragma unroll
                for (int u = 0; u < kL; ++u) { cur[u] = nxt[u]; curp[u] = nxtp[u]; }
            } else if (vn < nv) {
                fetch(vn, cur, curp);
            }
        }
        flush();
        if (st.m == INFINITY) { st.pinf = 1; st.m = -INFINITY; st.s = 0.0; }
        if ((n & 1) && tid == 0) {
            if (SCEN) {
                const double hv = h[n - 1];
                argmin_merge(amin_v, amin_i, amin_c, hv, n - 1, 1);
                unwritten = unwritten || (unsigned long long)__double_as_longlong(hv) == kUnwrittenBits;
                double x = c0 - hv;
                if (lnprior) x += lnprior[fin.dense ? (fin.branch ? fin.N - n : n - 1) : (long)src_idx[n - 1]];
                if (x < floor_x) x = -INFINITY;
                lme_fold4(st, x, -INFINITY, -INFINITY, -INFINITY);
            } else {
                lme_fold4(st, lme_value(logw, h, lnprior, c0, n - 1), -INFINITY, -INFINITY, -INFINITY);
            }
        }
    } else {
        for (long i0 = tid * 4; i0 < n; i0 += stride * 4) {
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                x[u] = (i0 + u < n) ? lme_value(logw, h, lnprior, c0, i0 + u) : -INFINITY;
            lme_fold4(st, x[0], x[1], x[2], x[3]);
        }
    }
    if (SCEN && unwritten) st.pinf |= 2;           // (bit 1 of the flag word travels with the partials: lme_merge ORs it)
    // wave combine (fixed butterfly order => deterministic)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Lme other;
        other.m = __shfl_xor(st.m, o, 64);
        other.s = __shfl_xor(st.s, o, 64);
        other.pinf = __shfl_xor(st.pinf, o, 64);
        lme_merge(st, other);
    }
    __shared__ double sm[4], ss[4];
    __shared__ int sp[4];
    __shared__ double av[4];
    __shared__ long ai[4], ac[4];
    const int wave = threadIdx.x >> 6;
    if (SCEN) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(amin_v, o, 64);
            const long oi = __shfl_xor(amin_i, o, 64);
            const long oc = __shfl_xor(amin_c, o, 64);
            argmin_merge(amin_v, amin_i, amin_c, ov, oi, oc);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sm[wave] = st.m; ss[wave] = st.s; sp[wave] = st.pinf;
        if (SCEN) { av[wave] = amin_v; ai[wave] = amin_i; ac[wave] = amin_c; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Lme t{sm[0], ss[0], sp[0]};
        for (int w = 1; w < 4; ++w) { Lme o{sm[w], ss[w], sp[w]}; lme_merge(t, o); }
        ws[3 * blockIdx.x + 0] = t.m;
        ws[3 * blockIdx.x + 1] = t.s;
        ws[3 * blockIdx.x + 2] = (double)t.pinf;
        if (SCEN) {
            double bv = av[0];
            long bi = ai[0], bc = ac[0];
            for (int w = 1; w < 4; ++w) argmin_merge(bv, bi, bc, av[w], ai[w], ac[w]);
            amin_pv[blockIdx.x] = bv;
            amin_pi[blockIdx.x] = bi;
            amin_pi[kLmeMaxBlocks + blockIdx.x] = bc;        // (the counts ride behind the positions)
        }
    }
    if (SCEN && fin.state) {
        // the block that finishes last turns the partials into the branch's record (scenario_final): what was a
        // launch of its own (final_kernel) until round 3
        __shared__ int is_last;
        if (threadIdx.x == 0) {
            __threadfence();                                   // this block's partials before its ticket
            const unsigned ticket = atomicAdd(&fin.state[0], 1u);
            is_last = ticket == nblocks - 1;
        }
        __syncthreads();
        if (is_last && threadIdx.x < 64) {
            __threadfence();                                   // the other blocks' partials after their tickets
            scenario_final(fin, ws, amin_pv, amin_pi, amin_pi + kLmeMaxBlocks, n, (int)threadIdx.x);
        }
    }
}

template <bool SCEN>
__global__ __launch_bounds__(256) void lme_partial_kernel(const double* __restrict__ logw,
                                                          const double* __restrict__ h,
                                                          const double* __restrict__ lnprior,
                                                          double c0, long n, int vec_ok,
                                                          double* __restrict__ ws,
                                                          const long* __restrict__ n_dev,
                                                          const int* __restrict__ src_idx,
                                                          double* __restrict__ amin_pv, long* __restrict__ amin_pi,
                                                          const double* __restrict__ bounds_base, const ScenFinal fin)
{
    lme_partial_body<SCEN>(logw, h, lnprior, c0, n, vec_ok, ws, n_dev, src_idx, amin_pv, amin_pi, bounds_base, fin);
}

// chain: the reductions of all branches in one launch, branch = blockIdx.y
struct LmeBranch {
    const double* h;
    const double* lnprior;
    double c0;
    double* ws;
    const long* n_dev;
    double* amin_pv;
    long* amin_pi;
    const double* bounds_base;
    ScenFinal fin;             // (fin.idx is the branch's src_idx)
};
struct LmeTab {
    LmeBranch b[kChainMaxBranches];
};
static_assert(sizeof(LmeTab) + 16 <= 4096, "kernel argument buffer");

__global__ __launch_bounds__(256) void lme_partial_kernel_star(LmeTab tab, long n_upper)
{
    const LmeBranch& b = tab.b[blockIdx.y];
    lme_partial_body<true>(nullptr, b.h, b.lnprior, b.c0, n_upper, 1, b.ws, b.n_dev, b.fin.idx, b.amin_pv, b.amin_pi,
                           b.bounds_base, b.fin);
}

__global__ __launch_bounds__(64) void lme_final_kernel(const double* __restrict__ ws, int nparts,
                                                       long n_total, double* __restrict__ out)
{
    // lane l folds partials l, l+64, ... in order, then a fixed butterfly: deterministic
    Lme t{-INFINITY, 0.0, 0};
    for (int i = threadIdx.x; i < nparts; i += 64) {
        Lme o{ws[3 * i], ws[3 * i + 1], ws[3 * i + 2] != 0.0};
        lme_merge(t, o);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Lme other;
        other.m = __shfl_xor(t.m, o, 64);
        other.s = __shfl_xor(t.s, o, 64);
        other.pinf = __shfl_xor(t.pinf, o, 64);
        lme_merge(t, other);
    }
    if (threadIdx.x != 0) return;
    double r;
    if (t.pinf) r = INFINITY;                                   // _numerics.py:46-47
    else if (t.m == -INFINITY) r = -INFINITY;                   // :49-50
    else r = log(t.s) + t.m - log((double)n_total);             // :51
    out[0] = r;
}

}  // namespace
