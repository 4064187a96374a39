// Developer microbenchmark: achievable HBM bandwidth for read-modify-write / read / write streams (double2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_rmw(double2* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = p[i]; v.x += 1.0; v.y -= 1.0; p[i] = v;
    }
}
__global__ void k_rd(const double2* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345) *out = s;
}
__global__ void k_wr(double2* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atol(argv[1]) : 256;
    size_t n = mb * 1024 * 1024 / 16;
    double2* p; double* o;
    hipMalloc(&p, n * 16); hipMalloc(&o, 8); hipMemset(p, 0, n * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_rmw, dim3(blocks), dim3(256), 0, 0, p, n);
                if (mode == 1) hipLaunchKernelGGL(k_rd, dim3(blocks), dim3(256), 0, 0, p, n, o);
                if (mode == 2) hipLaunchKernelGGL(k_wr, dim3(blocks), dim3(256), 0, 0, p, n);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            double bytes = (mode == 0 ? 2.0 : 1.0) * n * 16;
            printf("%zu MB blocks=%d %s: %.1f us  %.2f TB/s\n", mb, blocks, mode == 0 ? "rmw" : mode == 1 ? "read" : "write", best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
