// Developer microbenchmark (round 3): can fp64 VALU FMAs run NEXT TO v_mfma_f64_16x16x4_f64 on this chip?
//   (a) v_fma_f64 alone: NACC independent chains per lane, W waves per SIMD
//   (b) 8 waves per workgroup (2 per SIMD): waves 0-3 issue MFMAs, waves 4-7 VALU FMAs, each group timed on its own (wall_clock64, 100 MHz)
//   (c) one wave interleaving R VALU FMAs behind every MFMA
// The question behind it: the MFMA-bound kernels (flush, n_g^3 products) sit at the 47.5 TFLOP/s the matrix instruction sustains; if the
// vector ALU delivers its 78.6 TFLOP/s peak at the same time, a hybrid tile schedule could lift them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/valu_mfma_mix.hip -o scripts/micro/bin/valu_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template<int NACC>
__device__ __forceinline__ double valu_loop(int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.001 * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    return s;
}
template<int NACC>
__device__ __forceinline__ double mfma_loop(int iters, double a, double b) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d)(0.0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return s;
}

// mode 0: all waves VALU; 1: all waves MFMA; 2: waves 0..3 MFMA, 4..7 VALU (blockDim 512)
template<int MODE>
__global__ __launch_bounds__(512) void k_mix(double* out, int it_mfma, int it_valu, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9 * threadIdx.x;
    const bool do_mfma = MODE == 1 || (MODE == 2 && wave < 4);
    long long w0 = wall_clock64();
    double s;
    if (do_mfma) s = mfma_loop<8>(it_mfma, a, b);
    else s = valu_loop<16>(it_valu, a, b);
    long long w1 = wall_clock64();
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = w1 - w0;
}
// mode (c): every wave issues one MFMA followed by R independent VALU FMAs
template<int R>
__global__ __launch_bounds__(256) void k_interleave(double* out, int iters, long long* cyc) {
    const double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9 * threadIdx.x;
    v4d macc[4];
    double vacc[R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) macc[i] = (v4d)(0.0);
#pragma unroll
    for (int i = 0; i < (R > 0 ? R : 1); ++i) vacc[i] = 0.001 * i;
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            macc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc[i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) vacc[r] = __builtin_fma(a, vacc[r], b);
        }
    }
    long long w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
#pragma unroll
    for (int i = 0; i < (R > 0 ? R : 1); ++i) s += vacc[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = w1 - w0;
}

double* out; long long* cyc; hipEvent_t ea, eb;
template<int MODE> void run_mix(int wgs_per_cu, int nthreads, int it_mfma, int it_valu, const char* what) {
    const int grid = 256 * wgs_per_cu;
    float best = 1e9; long long hc[8];
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(cyc, 0, 64);
        hipEventRecord(ea);
        hipLaunchKernelGGL((k_mix<MODE>), dim3(grid), dim3(nthreads), 0, 0, out, it_mfma, it_valu, cyc);
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
        hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    }
    const int waves = nthreads / 64;
    const int n_m = MODE == 1 ? waves : (MODE == 2 ? 4 : 0), n_v = waves - n_m;
    // per-group rate from the group's own duration in workgroup 0 (10 ns ticks), scaled to the whole grid
    const double fl_m = 2048.0 * 8 * it_mfma * n_m * grid, fl_v = 2.0 * 64 * 16 * it_valu * n_v * grid;
    double t_m = 0, t_v = 0;
    for (int w = 0; w < waves; ++w) { const bool m = MODE == 1 || (MODE == 2 && w < 4); (m ? t_m : t_v) = std::max(m ? t_m : t_v, (double)hc[w] * 1e-8); }
    printf("%-44s %d wg/CU x %d waves: kernel %7.3f ms;", what, wgs_per_cu, waves, best);
    if (n_m) printf("  MFMA group %6.1f TFLOP/s over its %.3f ms;", fl_m / t_m / 1e12, t_m * 1e3);
    if (n_v) printf("  VALU group %6.1f TFLOP/s over its %.3f ms;", fl_v / t_v / 1e12, t_v * 1e3);
    printf("  total %6.1f TFLOP/s over the kernel\n", (fl_m + fl_v) / best / 1e9);
}
template<int R> void run_il(int wgs_per_cu, const char* what) {
    const int grid = 256 * wgs_per_cu, iters = 3000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(ea);
        hipLaunchKernelGGL((k_interleave<R>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
    }
    const double fm = 2048.0 * 4 * iters * 4.0 * grid, fv = 2.0 * 64 * R * 4 * iters * 4.0 * grid;
    printf("%-44s %d wg/CU: %7.3f ms; MFMA %5.1f + VALU %5.1f = %5.1f TFLOP/s\n", what, wgs_per_cu, best, fm / best / 1e9, fv / best / 1e9, (fm + fv) / best / 1e9);
}
int main() {
    hipMalloc(&out, 8 * 512 * 4096); hipMalloc(&cyc, 64);
    hipEventCreate(&ea); hipEventCreate(&eb);
    run_mix<0>(1, 256, 0, 4000, "(a) v_fma_f64 alone");
    run_mix<0>(2, 256, 0, 4000, "(a) v_fma_f64 alone");
    run_mix<0>(1, 512, 0, 4000, "(a) v_fma_f64 alone");
    run_mix<0>(2, 512, 0, 4000, "(a) v_fma_f64 alone");
    run_mix<1>(1, 256, 3000, 0, "    v_mfma_f64_16x16x4 alone");
    run_mix<1>(1, 512, 3000, 0, "    v_mfma_f64_16x16x4 alone");
    run_mix<2>(1, 512, 3000, 3000 * 8 * 6, "(b) 4 MFMA waves + 4 VALU waves per workgroup");
    run_mix<2>(1, 512, 3000, 3000 * 8 * 12, "(b) 4 MFMA waves + 4 VALU waves, more VALU");
    run_mix<2>(2, 512, 3000, 3000 * 8 * 6, "(b) 4 MFMA waves + 4 VALU waves per workgroup");
    run_il<0>(2, "(c) MFMA only (4 accumulators)");
    run_il<4>(2, "(c) 4 VALU FMAs behind every MFMA");
    run_il<8>(2, "(c) 8 VALU FMAs behind every MFMA");
    run_il<16>(2, "(c) 16 VALU FMAs behind every MFMA");
    run_il<24>(2, "(c) 24 VALU FMAs behind every MFMA");
    run_il<16>(1, "(c) 16 VALU FMAs behind every MFMA");
    return 0;
}
