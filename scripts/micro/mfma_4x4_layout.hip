// Developer probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 and the effect of cbsz / abid (A-block broadcast).
// A and B are given as functions of the lane that encode the lane itself; from D we read which (A lane, B lane) pairs were multiplied.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template<int CBSZ, int ABID>
__global__ void k(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, CBSZ, ABID, 0);
}
template<int CBSZ, int ABID> void probe(double* dA, double* dB, double* dD) {
    // pass p: A[lane] = (lane == a ? 1 : 0) for every a in turn is too slow to read; use two passes with A = 1 + lane, B = 64^...:
    // instead: for each a-lane one launch with A = e_a, B = 1 + lane  ->  D[l] = sum over k of [A lane a is (i,k) of l's block] * B(k, j)
    double hA[64], hB[64], hD[64];
    printf("cbsz %d abid %d: for each A lane a: the D lanes that receive it and the B lane it was multiplied with\n", CBSZ, ABID);
    for (int a = 0; a < 64; ++a) {
        for (int l = 0; l < 64; ++l) { hA[l] = (l == a) ? 1.0 : 0.0; hB[l] = 1.0 + l; }
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        printf("  A lane %2d ->", a);
        for (int l = 0; l < 64; ++l) if (hD[l] != 0.0) printf(" D%d*B%d", l, (int)std::lround(hD[l]) - 1);
        printf("\n");
    }
}
int main() {
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
    probe<0, 0>(dA, dB, dD);
    probe<2, 0>(dA, dB, dD);
    probe<2, 1>(dA, dB, dD);
    probe<2, 3>(dA, dB, dD);
    probe<1, 0>(dA, dB, dD);
    probe<1, 1>(dA, dB, dD);
    return 0;
}
