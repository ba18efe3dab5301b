// Does a VALU instruction cost less when part of the wave's lanes are masked off?  gfx950, wave64: a wave
// instruction is issued as 16-lane passes; this probe times a chain of independent v_fma_f64 with the
// lowest N lanes active (the shape of a partly filled trip of the pair loop: pairs sit in lanes 0..N-1).
// build: hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void probe(double* out, long long* cyc, int iters, int nactive)
{
    double b = 1.0000001, c = 1e-9;
    double r0 = 1.0 + threadIdx.x * 1e-9, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    const int lane = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
    if (lane < nactive) {
        for (int it = 0; it < iters; ++it) {
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int iters = 2000, blocks = 256 * 4, threads = 256;       // 4 waves per SIMD on every CU
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    printf("v_fma_f64, 4 independent chains, 4 waves per SIMD; wall time of the launch for the lowest N lanes active\n");
    for (int n : {64, 48, 33, 32, 17, 16, 8, 1}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, n);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double insts = (double)iters * 64;          // wave instructions per wave
        printf("  N = %2d: %.3f ms  = %.2f cycles per wave instruction per SIMD (at 2.4 GHz, 4 waves)\n", n, ms,
               ms * 1e-3 * 2.4e9 / (insts * 4));
    }
    return 0;
}
