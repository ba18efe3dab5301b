// Accuracy of the gfx950 fp64 seed instructions and issue cost of the fp32 alternatives.
// build: hipcc --offload-arch=gfx950 -O3 -o trans_probe trans_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void seeds(const double* x, double* rcp, double* rsq, double* sq, double* rcp32, double* rsq32, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], a, b, c;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(a) : "v"(v));
    asm volatile("v_rsq_f64 %0, %1" : "=v"(b) : "v"(v));
    asm volatile("v_sqrt_f64 %0, %1" : "=v"(c) : "v"(v));
    rcp[i] = a; rsq[i] = b; sq[i] = c;
    float f = (float)v, g, h;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(g) : "v"(f));
    asm volatile("v_rsq_f32 %0, %1" : "=v"(h) : "v"(f));
    rcp32[i] = (double)g; rsq32[i] = (double)h;
}

template <int OP>
__global__ void cost(double* out, int iters)
{
    double r0 = 1.0 + threadIdx.x * 1e-9;
    float f0 = 1.0f + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(f0));) }
        if (OP == 1) { REP16(asm volatile("v_rsq_f32 %0, %0" : "+v"(f0));) }
        if (OP == 2) { REP16(asm volatile("v_sqrt_f32 %0, %0" : "+v"(f0));) }
        if (OP == 3) { REP16(asm volatile("v_sqrt_f64 %0, %0" : "+v"(r0));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(r0));) }
        if (OP == 5) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r0) : "v"(f0));) }
        if (OP == 6) { REP16(asm volatile("v_cvt_f32_f64 %1, %0\n v_rcp_f32 %1, %1\n v_cvt_f64_f32 %0, %1" : "+v"(r0), "+v"(f0));) }
        if (OP == 7) { REP16(asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(r0));) }
        if (OP == 8) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0));) }
        if (OP == 9) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(r0));) }
        if (OP == 10) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(r0) : "vcc");) }
        if (OP == 11) { REP16(asm volatile("v_max_f64 %0, %0, %0" : "+v"(r0));) }
        if (OP == 12) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(r0));) }
        if (OP == 13) { REP16(asm volatile("v_mov_b64 %0, %0" : "+v"(r0));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + f0;
}

template <int OP>
void time_op(const char* name, int per_rep)
{
    double* out;
    hipMalloc(&out, sizeof(double) * 1024 * 512);
    const int iters = 2000;
    hipLaunchKernelGGL((cost<OP>), dim3(512), dim3(1024), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((cost<OP>), dim3(512), dim3(1024), 0, 0, out, iters);   // 8 waves per SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %6.2f clk per wave-instruction (2.4 GHz, 8 waves/SIMD)\n", name,
           ms * 1e6 / ((double)iters * 16 * per_rep * 8) * 2.4);
    hipFree(out);
}

int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30, 30);
    for (auto& v : x) v = std::exp2(u(g));
    double *dx, *d[5];
    hipMalloc(&dx, n * 8); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (auto& p : d) hipMalloc(&p, n * 8);
    hipLaunchKernelGGL(seeds, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], n);
    std::vector<double> h(n);
    const char* names[5] = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32(cvt)", "v_rsq_f32(cvt)"};
    for (int k = 0; k < 5; ++k) {
        hipMemcpy(h.data(), d[k], n * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ex = (k == 0 || k == 3) ? 1.0L / x[i] : (k == 2 ? sqrtl(x[i]) : 1.0L / sqrtl(x[i]));
            double e = fabs((double)((h[i] - ex) / ex));
            if (e > worst) worst = e;
        }
        printf("%-16s max rel err %.3e = 2^%.2f\n", names[k], worst, std::log2(worst));
    }
    time_op<0>("v_rcp_f32", 1); time_op<1>("v_rsq_f32", 1); time_op<2>("v_sqrt_f32", 1);
    time_op<3>("v_sqrt_f64", 1); time_op<9>("v_rcp_f64", 1);
    time_op<4>("v_cvt_f32_f64", 1); time_op<5>("v_cvt_f64_f32", 1);
    time_op<6>("cvt + v_rcp_f32 + cvt", 3);
    time_op<7>("v_ldexp_f64", 1); time_op<8>("v_fma_f32", 1); time_op<10>("v_cmp_lt_f64", 1);
    time_op<11>("v_max_f64", 1); time_op<12>("v_pk_fma_f32", 1); time_op<13>("v_mov_b64", 1);
    return 0;
}
