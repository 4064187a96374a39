// Developer microbenchmark (round 3): what does hipGraph replay buy for a chain of small DEPENDENT kernels on this ROCm (7.2)?
// N launches of a tiny kernel (one workgroup, ~2 us of work) on one stream: direct launches vs. the same sequence captured once and replayed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/graph_gap.hip -o scripts/micro/bin/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(double* x, int iters) {
    double v = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
    x[threadIdx.x] = v;
}
int main() {
    double* x; hipMalloc(&x, 256 * 8); hipMemset(x, 0, 256 * 8);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int iters : {50, 1000}) {
        for (int N : {2000, 8000}) {
            // direct
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, x, iters);
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            hipEventRecord(a, st);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, x, iters);
            hipEventRecord(b, st); hipEventSynchronize(b);
            double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("iters %4d N %5d  direct: %8.2f us per launch (device span), wall %8.2f us per launch\n", iters, N, 1e3 * ms / N, 1e6 * wall / N);
            // graph
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, x, iters);
            hipStreamEndCapture(st, &g);
            auto c0 = std::chrono::steady_clock::now();
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            double tinst = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);
            t0 = std::chrono::steady_clock::now();
            hipEventRecord(a, st);
            hipGraphLaunch(ge, st);
            hipEventRecord(b, st); hipEventSynchronize(b);
            wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            hipEventElapsedTime(&ms, a, b);
            printf("iters %4d N %5d  graph : %8.2f us per launch (device span), wall %8.2f us per launch, instantiate %.1f ms\n", iters, N, 1e3 * ms / N, 1e6 * wall / N, 1e3 * tinst);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
