// Developer microbenchmark: which part of the flush kernel costs what (32 chains, n = 512, K = 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template<int MODE>   // bit0: operand loads, bit1: mfma, bit2: read K from memory, bit3: C loaded late
__global__ __launch_bounds__(256) void k_flush(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                               cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev, size_t cs) {
    X += blockIdx.z * cs; Gr += blockIdx.z * cs; G += blockIdx.z * cs;
    int K = Kmax;
    if (MODE & 4) { int kd = Kdev[blockIdx.z * 64]; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.x * 64 + (wave >> 1) * 32, j0 = blockIdx.y * 64 + (wave & 1) * 32;
    cplx c[2][2][4];
    if (!(MODE & 8)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[a][b][r] = G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15];
    }
    v4d acc_re[2][2], acc_im[2][2], acc_p3[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p3[a][b] = (v4d)(0.0); }
    if (MODE & 1) {
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int gk = k0 + l4;
        cplx af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = X[(size_t)gk * ldx + i0 + a * 16 + l15];
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = Gr[(size_t)(j0 + b * 16 + l15) * ldg + gk];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (MODE & 16) {
                // Gauss: P1 = br ar, P2 = bi ai, P3 = (br + bi)(ar + ai): re = P1 - P2, im = P3 - P1 - P2
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].y, acc_im[a][b], 0, 0, 0);
                acc_p3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x + bf[b].y, af[a].x + af[a].y, acc_p3[a][b], 0, 0, 0);
                } else if (MODE & 2) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bf[b].y, af[a].y, acc_re[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].x, acc_im[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].y, acc_im[a][b], 0, 0, 0);
                } else { acc_re[a][b][0] += af[a].x * bf[b].x; acc_im[a][b][0] += af[a].y * bf[b].y; }
            }
    }
    }
    if (MODE & 8) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[a][b][r] = G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                G[(size_t)(j0 + b * 16 + l4 + 4 * r) * ldc + i0 + a * 16 + l15] = (MODE & 16)
                    ? make_double2(c[a][b][r].x + (acc_re[a][b][r] - acc_im[a][b][r]), c[a][b][r].y + (acc_p3[a][b][r] - acc_re[a][b][r] - acc_im[a][b][r]))
                    : make_double2(c[a][b][r].x + acc_re[a][b][r], c[a][b][r].y + acc_im[a][b][r]);
}
template<int MODE> float run(cplx* X, cplx* Gr, cplx* G, int* Kd, int n, int nb, size_t cs, hipEvent_t a, hipEvent_t b) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_flush<MODE>), dim3(n / 64, n / 64, nb), dim3(256), 0, 0, X, n, Gr, 32, G, n, n, 32, Kd, cs);
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
    int hk[64 * 32]; for (int i = 0; i < 64 * 32; ++i) hk[i] = 32;
    CK(hipMemcpy(Kd, hk, sizeof(hk), hipMemcpyHostToDevice));
    cplx *G = p, *X = p + (size_t)n * n, *Gr = X + (size_t)n * 32;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("rmw only            : %.1f us\n", run<0>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("+ Kdev              : %.1f us\n", run<4>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("+ operand loads     : %.1f us\n", run<1>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("+ operands + mfma   : %.1f us\n", run<3>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("full (Kdev)         : %.1f us\n", run<7>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("full, C loaded late : %.1f us\n", run<15>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("gauss 3-mult        : %.1f us\n", run<1 + 4 + 16>(X, Gr, G, Kd, n, nb, cs, a, b));
    printf("gauss, C late       : %.1f us\n", run<1 + 4 + 8 + 16>(X, Gr, G, Kd, n, nb, cs, a, b));
    return 0;
}
