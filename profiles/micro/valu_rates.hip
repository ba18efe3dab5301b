// VALU issue-rate probe for gfx950: cycles per wave-instruction for the opcodes the light-curve
// kernel is made of, at 1..4 waves per SIMD, dependent chain vs 4 independent chains.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP, int ILP>
__global__ void probe(double* out, long long* cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    double r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3;
    int i0 = threadIdx.x, i1 = 3;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // v_fma_f64
            if (ILP == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r0) : "v"(b), "v"(c));) }
            else { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));) }
        } else if (OP == 1) {  // v_mul_f64
            if (ILP == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r0) : "v"(b));) }
            else { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b));) }
        } else if (OP == 2) {  // v_add_f64
            if (ILP == 1) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(r0) : "v"(c));) }
            else { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));) }
        } else if (OP == 3) {  // v_rcp_f64
            if (ILP == 1) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(r0));) }
            else { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
        } else if (OP == 4) {  // v_rsq_f64
            if (ILP == 1) { REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(r0));) }
            else { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
        } else if (OP == 5) {  // v_mov_b32
            if (ILP == 1) { REP16(asm volatile("v_mov_b32 %0, %0" : "+v"(i0));) }
            else { REP16(asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %0, %0\n v_mov_b32 %1, %1" : "+v"(i0), "+v"(i1));) }
        } else if (OP == 6) {  // v_cndmask_b32 (vcc)
            if (ILP == 1) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i1) : "vcc");) }
            else { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(i0), "+v"(i1) : : "vcc");) }
        } else if (OP == 7) {  // v_fma_f64 with an SGPR addend
            if (ILP == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r0) : "v"(b), "s"(c));) }
            else { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "s"(c));) }
        } else if (OP == 8) {  // v_fma_f64 interleaved with s_mov_b32 pairs (literal constants in SGPRs)
            REP16(asm volatile("s_mov_b32 s40, 0x11111111\n s_mov_b32 s41, 0x3f811111\n v_fma_f64 %0, %0, %1, s[40:41]" : "+v"(r0) : "v"(b) : "s40", "s41");)
        } else if (OP == 9) {  // v_fmac_f64 preceded by two literal v_mov (what the compiler emits for Horner)
            REP16(asm volatile("v_mov_b32 %1, 0x11111111\n v_mov_b32 %2, 0x3f811111\n v_fma_f64 %0, %0, %3, %0" : "+v"(r0), "+v"(i0), "+v"(i1) : "v"(b));)
        } else if (OP == 10) {  // v_cmp_lt_f64 + v_cndmask pair
            REP16(asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i1), "v"(r0), "v"(b) : "vcc");)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + i0 + i1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int ILP>
void run(const char* name, int per_rep)
{
    double* out; long long* cyc;
    const int iters = 2000;
    hipMalloc(&out, sizeof(double) * 256 * 16 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 16 * 1024);
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        // 256 CUs x 4 SIMDs: one block = 64 * 4 * waves_per_simd threads, one block per CU
        const int threads = 64 * 4 * waves_per_simd;
        const int blocks = 256;
        if (threads > 1024) {
            // two blocks of 1024 per CU
        }
        const int bt = threads > 1024 ? 1024 : threads;
        const int nb = threads > 1024 ? blocks * (threads / 1024) : blocks;
        hipLaunchKernelGGL((probe<OP, ILP>), dim3(nb), dim3(bt), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<OP, ILP>), dim3(nb), dim3(bt), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(nb);
        hipMemcpy(h.data(), cyc, sizeof(long long) * nb, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= nb;
        const double insts = (double)iters * 16 * per_rep;          // per wave
        // shader clock from the event time: cycles = ms * f; report ns per wave-instruction per SIMD
        const double ns_per_inst_simd = ms * 1e6 / (insts * waves_per_simd);
        printf("%-28s ILP%d waves/SIMD %d  %7.3f ms  %6.2f ns/inst/SIMD (x2.4 = %5.2f clk)  s_memtime ticks/inst/wave %6.2f\n",
               name, ILP, waves_per_simd, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4, mean / insts);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 1>("v_fma_f64", 1);  run<0, 4>("v_fma_f64", 4);
    run<1, 1>("v_mul_f64", 1);  run<1, 4>("v_mul_f64", 4);
    run<2, 1>("v_add_f64", 1);  run<2, 4>("v_add_f64", 4);
    run<3, 1>("v_rcp_f64", 1);  run<3, 4>("v_rcp_f64", 4);
    run<4, 1>("v_rsq_f64", 1);  run<4, 4>("v_rsq_f64", 4);
    run<5, 1>("v_mov_b32", 1);  run<5, 4>("v_mov_b32", 4);
    run<6, 1>("v_cndmask_b32", 1); run<6, 4>("v_cndmask_b32", 4);
    run<7, 1>("v_fma_f64 sgpr addend", 1); run<7, 4>("v_fma_f64 sgpr addend", 4);
    run<8, 1>("2 s_mov + v_fma_f64", 1);
    run<9, 1>("2 v_mov lit + v_fma_f64", 3);
    run<10, 1>("v_cmp_f64 + v_cndmask", 2);
    return 0;
}
