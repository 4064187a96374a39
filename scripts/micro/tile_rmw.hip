// Developer microbenchmark: RMW bandwidth over nb column-major n x n complex matrices for different tilings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// the access pattern of k_flush: workgroup = 64 x 64 tile, wave = 32 x 32, lane (l15, l4) -> rows l15 (+16), cols l4 + 4r (+16)
__global__ __launch_bounds__(256) void k_tile64(double2* G, int n, size_t stride) {
    G += blockIdx.z * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.x * 64 + (wave >> 1) * 32, j0 = blockIdx.y * 64 + (wave & 1) * 32;
    double2 c[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[a][b][r] = G[(size_t)(j0 + b * 16 + l4 + 4 * r) * n + i0 + a * 16 + l15];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) { double2 v = c[a][b][r]; v.x += 1.0; G[(size_t)(j0 + b * 16 + l4 + 4 * r) * n + i0 + a * 16 + l15] = v; }
}
// workgroup = CW full columns (contiguous n*CW*16 bytes), threads stream it linearly
template<int CW>
__global__ __launch_bounds__(256) void k_panel(double2* G, int n, size_t stride) {
    G += blockIdx.z * stride + (size_t)blockIdx.x * CW * n;
    const int total = CW * n;
    double2 v[8];
    for (int base = 0; base < total; base += 256 * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { int idx = base + u * 256 + threadIdx.x; if (idx < total) v[u] = G[idx]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { int idx = base + u * 256 + threadIdx.x; if (idx < total) { v[u].x += 1.0; G[idx] = v[u]; } }
    }
}
int main(int argc, char** argv) {
    int n = 512, nb = argc > 1 ? atoi(argv[1]) : 32;
    size_t stride = argc > 2 ? (size_t)atol(argv[2]) * 1024 * 1024 / 16 : (size_t)n * n;   // elements between matrices
    double2* p;
    CK(hipMalloc(&p, stride * nb * 16)); CK(hipMemset(p, 0, stride * nb * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double bytes = 2.0 * n * n * 16 * nb;
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(k_tile64, dim3(n / 64, n / 64, nb), dim3(256), 0, 0, p, n, stride);
            if (mode == 1) hipLaunchKernelGGL(k_panel<8>, dim3(n / 8, 1, nb), dim3(256), 0, 0, p, n, stride);
            if (mode == 2) hipLaunchKernelGGL(k_panel<16>, dim3(n / 16, 1, nb), dim3(256), 0, 0, p, n, stride);
            if (mode == 3) hipLaunchKernelGGL(k_panel<32>, dim3(n / 32, 1, nb), dim3(256), 0, 0, p, n, stride);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const char* nm[] = {"tile64x64", "panel8", "panel16", "panel32"};
        printf("nb=%d stride=%zuMB %s: %.1f us  %.2f TB/s\n", nb, stride * 16 >> 20, nm[mode], best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
