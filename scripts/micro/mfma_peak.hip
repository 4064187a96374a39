// Developer microbenchmark: what v_mfma_f64_16x16x4_f64 sustains on this box with nothing else going on (no memory traffic):
// NACC independent accumulators per wave, W waves per SIMD; clock64() ticks per MFMA seen by wave 0 (s_memtime: shader clock).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template<int NACC, bool ZERO, bool SMALL>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, long long* cyc) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d)(0.0);
    double a = ZERO ? 0.0 : 1.0 + threadIdx.x * 1e-9, b = ZERO ? 0.0 : 1.0 - threadIdx.x * 1e-9;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SMALL) acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
// the pattern a 16x16 tile needs: one k-step = 4 MFMAs whose second operand is the same register rotated by 0/4/8/12 lanes (DPP);
// NT independent tiles per wave, fresh operand values every step
template<int S> __device__ __forceinline__ double row_rot(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + 4 * S, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + 4 * S, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template<int NT, bool ROT>
__global__ __launch_bounds__(256) void k_mfma_rot(double* out, int iters, long long* cyc) {
    double acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[i][s] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const double bb = b + i;                       // a new operand per tile and step
            const double b1 = ROT ? row_rot<1>(bb) : bb + 1.0, b2 = ROT ? row_rot<2>(bb) : bb + 2.0, b3 = ROT ? row_rot<3>(bb) : bb + 3.0;
            acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc[i][1], 0, 0, 0);
            acc[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b2, acc[i][2], 0, 0, 0);
            acc[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b3, acc[i][3], 0, 0, 0);
        }
        b += 1e-9;
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = 0; }
}
double* out; long long* cyc; hipEvent_t ea, eb;
template<int NT, bool ROT> void run_rot(int wgs_per_cu, const char* what) {
    const int iters = 6000 / NT;
    const int grid = 256 * wgs_per_cu;
    float best = 1e9; long long hc[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(ea);
        hipLaunchKernelGGL((k_mfma_rot<NT, ROT>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
        hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    }
    const double flops = 512.0 * 4 * NT * iters * 4.0 * grid;
    printf("%-34s NT %d, %d wave(s)/SIMD: %7.3f ms, %5.1f TFLOP/s; wave 0: %.1f ticks per MFMA\n", what, NT, wgs_per_cu, best, flops / best / 1e9,
           (double)hc[0] / (4.0 * NT * iters));
}
template<int NACC, bool ZERO, bool SMALL> void run(int wgs_per_cu, const char* what) {
    const int iters = 24000 / NACC;
    const int grid = 256 * wgs_per_cu;
    float best = 1e9; long long hc[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(ea);
        hipLaunchKernelGGL((k_mfma<NACC, ZERO, SMALL>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
        hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    }
    const double per = SMALL ? 512.0 : 2048.0;
    const double flops = per * NACC * iters * 4.0 * grid;
    printf("%-22s NACC %2d, %d wave(s)/SIMD: %7.3f ms, %5.1f TFLOP/s; wave 0: %.1f clock64 ticks per MFMA, shader clock %.2f GHz (wall_clock64 at 100 MHz)\n",
           what, NACC, wgs_per_cu, best, flops / best / 1e9, (double)hc[0] / ((double)NACC * iters), (double)hc[0] / ((double)hc[1] * 10.0) );
}
int main() {
    hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&cyc, 16);
    hipEventCreate(&ea); hipEventCreate(&eb);
    run<1, false, false>(1, "16x16x4 dependent");
    run<2, false, false>(1, "16x16x4");
    run<4, false, false>(1, "16x16x4");
    run<12, false, false>(1, "16x16x4");
    run<12, false, false>(2, "16x16x4");
    run<12, false, false>(3, "16x16x4");
    run<12, false, false>(4, "16x16x4");
    run<4, false, false>(8, "16x16x4");
    run<12, true, false>(2, "16x16x4 zero operands");
    run<1, false, true>(1, "4x4x4 dependent");
    run<12, false, true>(1, "4x4x4");
    run<12, false, true>(2, "4x4x4");
    run<12, false, true>(4, "4x4x4");
    run_rot<3, false>(2, "4x4x4, operands from VALU adds");
    run_rot<3, true>(1, "4x4x4, operands rotated by DPP");
    run_rot<3, true>(2, "4x4x4, operands rotated by DPP");
    run_rot<3, true>(3, "4x4x4, operands rotated by DPP");
    run_rot<6, true>(2, "4x4x4, operands rotated by DPP");
    // long run: does the rate hold (power)?
    hipEventRecord(ea);
    for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL((k_mfma<12, false, false>), dim3(512), dim3(256), 0, 0, out, 4000, cyc);
    hipEventRecord(eb); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    printf("200 launches back to back (2 waves per SIMD): %.1f ms, %.1f TFLOP/s sustained\n", ms, 2048.0 * 12 * 4000 * 4.0 * 512 * 200 / ms / 1e9);
    return 0;
}
