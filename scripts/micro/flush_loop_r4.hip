// Developer microbenchmark (round 4): what the MFMA loop of the flush sustains on its own -- operand fragments from LDS, no global traffic
// inside the loop -- as a function of resident waves per SIMD, for the 3M (3 MFMAs + 4 fp64 adds per complex 16x16x4 step) and the 4M form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/flush_loop_r4.hip -o /tmp/flush_loop_r4 && /tmp/flush_loop_r4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template<bool M3, int WAVES, int MINB, int SRC>      // SRC 0: fragments from LDS; 1: fragments constant in registers (pure issue rate)
__global__ __launch_bounds__(64 * WAVES, MINB) void k_loop(cplx* __restrict__ out, int ktot, int reps) {
    extern __shared__ cplx sm[];                 // Xs[32][64], Gs[32][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int i = tid; i < 2 * 32 * 64; i += 64 * WAVES) sm[i] = make_double2(1e-3 * (i % 17), 1e-3 * (i % 13));
    __syncthreads();
    const cplx* Xs = sm; const cplx* Gs = sm + 32 * 64;
    const int wi = ((wave >> 1) & 1) * 32, wj = (wave & 1) * 32;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    auto loadf = [&](int kl, cplx (&f)[4]) {
        if (SRC == 1) return;
        const cplx* xr = Xs + ((kl & 31) + l4) % 32 * 64 + wi + l15;
        const cplx* gr = Gs + ((kl & 31) + l4) % 32 * 64 + wj + l15;
        f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
    };
    auto mac = [&](const cplx (&f)[4]) {
        if (M3) {
            double asum[2], bsum[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f[2 + b].y, f[a].y, acc_re[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].x, acc_im[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].y, acc_im[a][b], 0, 0, 0);
                }
        }
    };
    cplx s[4], t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s[q] = make_double2(1e-3 * lane, 1e-3 * q); t[q] = make_double2(2e-3 * lane, 1e-3 * q); }
    for (int r = 0; r < reps; ++r) {
        loadf(0, s);
        for (int k0 = 0; k0 < ktot; k0 += 8) {
            loadf(k0 + 4, t);
            __builtin_amdgcn_sched_barrier(0);
            mac(s);
            __builtin_amdgcn_sched_barrier(0);
            loadf(k0 + 8, s);
            __builtin_amdgcn_sched_barrier(0);
            mac(t);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    cplx* o = out + ((size_t)blockIdx.x * 64 * WAVES + tid) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            v4d re = M3 ? (acc_re[a][b] - acc_p2[a][b]) : acc_re[a][b];
            o[a * 2 + b] = make_double2(re[0] + re[1] + re[2] + re[3], acc_im[a][b][0] + acc_im[a][b][1] + acc_im[a][b][2] + acc_im[a][b][3]);
        }
}

template<class F> static float timeit(hipEvent_t a, hipEvent_t b, F f) {
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0); f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f;
}
int main() {
    cplx* out; CK(hipMalloc(&out, (size_t)4096 * 512 * 4 * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int ktot = 64, reps = 200;
    const size_t lds = 2 * 32 * 64 * 16;     // 64 KB
#define RUN(M3, WAVES, MINB, SRC, WGPERCU, LDS, name) do { \
        CK(hipFuncSetAttribute((const void*)k_loop<M3, WAVES, MINB, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
        const int grid = 256 * WGPERCU; \
        float us = timeit(a, b, [&] { hipLaunchKernelGGL((k_loop<M3, WAVES, MINB, SRC>), dim3(grid), dim3(64 * WAVES), LDS, 0, out, ktot, reps); }); \
        const double steps = (double)grid * WAVES * reps * (ktot / 4);  /* complex 32x32x4 steps per wave */ \
        const double mfma = steps * (M3 ? 12 : 16), flop_alg = steps * 32 * 32 * 4 * 8; \
        printf("%-58s %9.1f us  %6.1f TFLOP/s executed (%.2f of 78.6), %6.1f algorithmic (8 flop per complex mac)\n", name, us, mfma * 2048 / us / 1e6, mfma * 2048 / us / 1e6 / 78.6, flop_alg / us / 1e6); \
    } while (0)
    RUN(true, 4, 1, 0, 1, lds, "3M, LDS fragments, 1 wave/SIMD");
    RUN(true, 4, 2, 0, 2, lds, "3M, LDS fragments, 2 waves/SIMD");
    RUN(true, 4, 2, 0, 2, lds / 2, "3M, LDS fragments, 2 waves/SIMD (grid 2/CU, 32 KB)");
    RUN(true, 8, 1, 0, 1, lds, "3M, LDS fragments, 8-wave WG = 2 waves/SIMD");
    RUN(true, 4, 3, 0, 3, lds / 2, "3M, LDS fragments, 3 waves/SIMD");
    RUN(true, 4, 1, 1, 1, lds, "3M, register fragments (issue rate), 1 wave/SIMD");
    RUN(true, 4, 2, 1, 2, lds, "3M, register fragments (issue rate), 2 waves/SIMD");
    RUN(false, 4, 1, 0, 1, lds, "4M, LDS fragments, 1 wave/SIMD");
    RUN(false, 4, 2, 0, 2, lds, "4M, LDS fragments, 2 waves/SIMD");
    RUN(false, 4, 3, 0, 3, lds / 2, "4M, LDS fragments, 3 waves/SIMD");
    RUN(false, 4, 1, 1, 1, lds, "4M, register fragments (issue rate), 1 wave/SIMD");
    RUN(false, 4, 2, 1, 2, lds, "4M, register fragments (issue rate), 2 waves/SIMD");
    return 0;
}
