// Developer microbenchmark: flush kernel, wave tile (16*TM) x 32, occupancy hints, C loaded late.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template<int TM, int MINB, int UNR, int EARLYC>
__global__ __launch_bounds__(256, MINB) void k_flush(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                               cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev, size_t cs) {
    X += blockIdx.z * cs; Gr += blockIdx.z * cs; G += blockIdx.z * cs;
    int K = Kmax;
    { int kd = Kdev[blockIdx.z * 64]; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.x * (32 * TM) + (wave >> 1) * (16 * TM), j0 = blockIdx.y * 64 + (wave & 1) * 32;
    v4d acc_re[TM][2], acc_im[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); }
    // EARLYC: odd workgroups start from the tile of G (accumulators initialised with it) so that on a CU the memory
    // phase of one workgroup meets the MFMA phase of another
    const bool early = EARLYC && ((blockIdx.x + blockIdx.y) & 1);
    if (early) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cplx c = G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15];
                    acc_re[a][b][r] = c.x; acc_im[a][b][r] = c.y;
                }
    }
#pragma unroll UNR
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int gk = k0 + l4;
        cplx af[TM], bf[2];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = X[(size_t)gk * ldx + i0 + a * 16 + l15];
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = Gr[(size_t)(j0 + b * 16 + l15) * ldg + gk];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bf[b].y, af[a].y, acc_re[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].x, acc_im[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].y, acc_im[a][b], 0, 0, 0);
            }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            cplx c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = early ? make_double2(0.0, 0.0) : G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15] = make_double2(c[r].x + acc_re[a][b][r], c[r].y + acc_im[a][b][r]);
        }
}
template<int TM, int MINB, int UNR, int EARLYC = 0> float run(cplx* X, cplx* Gr, cplx* G, int* Kd, int n, int nb, size_t cs, hipEvent_t a, hipEvent_t b, int K = 32) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_flush<TM, MINB, UNR, EARLYC>), dim3(n / (32 * TM), n / 64, nb), dim3(256), 0, 0, X, n, Gr, K, G, n, n, K, Kd, cs);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best * 1e3f;
}
int main(int argc, char** argv) {
    int n = 512, nb = 32;
    size_t cs = (size_t)240 * 1024 * 1024 / 16;
    cplx* p; int* Kd;
    CK(hipMalloc(&p, cs * nb * 16)); CK(hipMemset(p, 0, cs * nb * 16));
    CK(hipMalloc(&Kd, 64 * nb * 4));
    int hk[64 * 32]; for (int i = 0; i < 64 * 32; ++i) hk[i] = 64;
    CK(hipMemcpy(Kd, hk, sizeof(hk), hipMemcpyHostToDevice));
    cplx *G = p, *X = p + (size_t)n * n, *Gr = X + (size_t)n * 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("K=64: TM=2 minb=2 unr=2 (current): %.1f us\n", run<2, 2, 2>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=2 minb=2 unr=4          : %.1f us\n", run<2, 2, 4>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=2 minb=2 unr=2 earlyC   : %.1f us\n", run<2, 2, 2, 1>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=2 minb=3 unr=2 earlyC   : %.1f us\n", run<2, 3, 2, 1>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=1 minb=4 unr=2          : %.1f us\n", run<1, 4, 2>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=1 minb=4 unr=2 earlyC   : %.1f us\n", run<1, 4, 2, 1>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("K=64: TM=1 minb=6 unr=4          : %.1f us\n", run<1, 6, 4>(X, Gr, G, Kd, n, nb, cs, a, b, 64));
    printf("TM=2 minb=1 unr=4: %.1f us\n", run<2, 1, 4>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=2 minb=2 unr=2: %.1f us\n", run<2, 2, 2>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=2 minb=3 unr=2: %.1f us\n", run<2, 3, 2>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=2 minb=4 unr=1: %.1f us\n", run<2, 4, 1>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=1 minb=2 unr=4: %.1f us\n", run<1, 2, 4>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=1 minb=4 unr=2: %.1f us\n", run<1, 4, 2>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=1 minb=6 unr=2: %.1f us\n", run<1, 6, 2>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("TM=1 minb=8 unr=1: %.1f us\n", run<1, 8, 1>(X, Gr, G, Kd, n, nb, cs, a, b));
    return 0;
}
