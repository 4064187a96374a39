// Developer microbenchmark: what v_mfma_f64_16x16x4_f64 sustains on this box with nothing else going on (no memory traffic):
// NACC independent accumulators per wave, W waves per SIMD.   Also reports the clock the waves saw (s_memrealtime vs s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template<int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, long long* cyc) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d)(0.0);
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        float best = 1e9; long long hc = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_mfma<12>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        }
        const double flops = 2048.0 * 12 * iters * 4.0 * grid;
        printf("%d workgroup(s) of 4 waves per CU: %.3f ms, %.1f TFLOP/s; wave 0: %lld clock64 ticks for %d MFMAs (%.1f per MFMA)\n",
               wgs_per_cu, best, flops / best / 1e9, hc, 12 * iters, (double)hc / (12.0 * iters));
    }
    // long run: does the rate hold (power)?
    hipEventRecord(a);
    for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(k_mfma<12>, dim3(512), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("200 launches back to back (2 workgroups per CU): %.1f ms, %.1f TFLOP/s sustained\n", ms, 2048.0 * 12 * iters * 4.0 * 512 * 200 / ms / 1e9);
    return 0;
}
