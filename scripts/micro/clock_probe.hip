// Developer probe (round 3): the shader clock the chip HOLDS while another process runs the DQMC sweep on it.
// One wave per XCD-sized group of workgroups samples s_memtime (shader-clock cycles) against wall_clock64 (100 MHz constant clock)
// over windows of ~1 ms for `seconds` seconds; the host prints min / mean / max MHz over all windows and workgroups.
// bench.py prices MFMA-bound kernels against 78.6 TFLOP/s = 256 CUs x 128 flop/clk x 2.4 GHz; if the chip holds less than 2.4 GHz under
// the fp64-MFMA + HBM load of the sweep, the roofline fractions understate how close the kernels are to what the silicon delivers.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/clock_probe.hip -o scripts/micro/bin/clock_probe
//   scripts/micro/bin/clock_probe 20 &  python bench.py --inprocess --batch 128 --sub-batches 1 --steps 10 --warmup 5 --no-cpu-baseline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void k_probe(double* mhz, int nwin, long long win_ticks) {
    if (threadIdx.x != 0) return;
    for (int w = 0; w < nwin; ++w) {
        const long long w0 = wall_clock64();
        const unsigned long long c0 = __builtin_readcyclecounter();
        long long w1;
        do { __builtin_amdgcn_s_sleep(64); w1 = wall_clock64(); } while (w1 - w0 < win_ticks);
        const unsigned long long c1 = __builtin_readcyclecounter();
        mhz[(size_t)blockIdx.x * nwin + w] = (double)(c1 - c0) / ((double)(w1 - w0) * 0.01);      // cycles per microsecond = MHz
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int nblk = 64, nwin = (int)(seconds * 1000.0);
    double* d;
    hipMalloc(&d, sizeof(double) * nblk * nwin);
    hipLaunchKernelGGL(k_probe, dim3(nblk), dim3(64), 0, 0, d, nwin, 100000LL);     // 1 ms windows at 100 MHz
    hipDeviceSynchronize();
    std::vector<double> h((size_t)nblk * nwin);
    hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // per window: mean over the probing workgroups; report the distribution over windows
    std::vector<double> win(nwin);
    for (int w = 0; w < nwin; ++w) { double s = 0; for (int b = 0; b < nblk; ++b) s += h[(size_t)b * nwin + w]; win[w] = s / nblk; }
    std::vector<double> sorted = win;
    std::sort(sorted.begin(), sorted.end());
    double mean = 0; for (double v : win) mean += v; mean /= nwin;
    printf("CLOCK_PROBE windows=%d of 1 ms  shader clock MHz: min %.0f  p10 %.0f  median %.0f  mean %.0f  p90 %.0f  max %.0f\n", nwin, sorted.front(),
           sorted[nwin / 10], sorted[nwin / 2], mean, sorted[(size_t)nwin * 9 / 10], sorted.back());
    // coarse time line: mean MHz per 500 ms
    for (int w = 0; w + 500 <= nwin; w += 500) { double s = 0; for (int i = 0; i < 500; ++i) s += win[w + i]; printf("  t=%4.1fs %.0f MHz\n", w / 1000.0, s / 500); }
    return 0;
}
