// Developer microbenchmark (round 2): complex-fp64 GEMM C = A B (n = 512, 128 chains) on v_mfma_f64_4x4x4_4b_f64 with the four
// blocks of the instruction used as four K-GROUPS of one 4 x 4 block of C (3M complex product: three accumulators per block).
// Workgroup = 8 waves = a 64 x 32 tile of C, wave (wm, wn) its 16 x 16 part; K in chunks of BK through LDS (double buffered).
// Production k_zgemm<2,2> (v_mfma_f64_16x16x4_f64, 64 x 64 tiles): 1911 us per launch of this shape = 62.6 TFLOP/s (8 flop per
// complex multiply-add).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/gemm_kq.hip -o /tmp/gemm_kq && /tmp/gemm_kq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <complex>
typedef double2 cplx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void chain_tile(int tiles, int nb, int& chain, int& tile) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    chain = (slot / tiles) * 8 + xcd;
    tile = slot % tiles;
}
template<int N> __device__ __forceinline__ double row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template<int BK>
__global__ __launch_bounds__(512, 2) void k_gemm_kq(const cplx* __restrict__ A, const cplx* __restrict__ B, cplx* __restrict__ C, int n, size_t cs, int nb) {
    constexpr int PA = 65, PB = 33;             // odd pitches: the four k-rows a 16-lane group reads fall on different banks
    __shared__ cplx sA[2][BK][PA];              // [k][row]
    __shared__ cplx sB[2][BK][PB];              // [k][col]
    const int tm = n / 64, tn = n / 32;
    int chain, tile;
    chain_tile(tm * tn, nb, chain, tile);
    A += chain * cs; B += chain * cs; C += chain * cs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                      // 4 x 2 waves
    const int i0 = (tile % tm) * 64, j0 = (tile / tm) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int jj = lane & 3, q = (lane >> 2) & 3;
    double p1[4][4], p2[4][4], p3[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { p1[r][c] = 0.0; p2[r][c] = 0.0; p3[r][c] = 0.0; }
    // staging: A chunk 64 rows x BK (BK * 64 / 512 elements per thread), B chunk BK x 32 (BK * 32 / 512 per thread)
    constexpr int NA = BK * 64 / 512, NB = BK * 32 / 512;
    cplx sta[NA], stb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) { const int idx = tid + 512 * e; sta[e] = A[(size_t)(k0 + idx / 64) * n + i0 + idx % 64]; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { const int idx = tid + 512 * e; stb[e] = B[(size_t)(j0 + idx / BK) * n + k0 + idx % BK]; }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NA; ++e) { const int idx = tid + 512 * e; sA[buf][idx / 64][idx % 64] = sta[e]; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { const int idx = tid + 512 * e; sB[buf][idx % BK][idx / BK] = stb[e]; }
    };
    gload(0); sstore(0);
    __syncthreads();
    for (int k0 = 0, buf = 0; k0 < n; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < n;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int kc = 0; kc < BK; kc += 16) {
            cplx xf[4], gf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) xf[r] = sA[buf][kc + 4 * q + l4][16 * wm + 4 * r + jj];      // second operand: rows of C
#pragma unroll
            for (int c = 0; c < 4; ++c) gf[c] = sB[buf][kc + 4 * q + l4][16 * wn + 4 * c + jj];      // first operand: columns of C
            double xs[4], gs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { xs[r] = xf[r].x + xf[r].y; gs[r] = gf[r].x + gf[r].y; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    p1[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(gf[c].x, xf[r].x, p1[r][c], 0, 0, 0);
                    p2[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(gf[c].y, xf[r].y, p2[r][c], 0, 0, 0);
                    p3[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(gs[c], xs[r], p3[r][c], 0, 0, 0);
                }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
    }
    cplx* cbase = C + (size_t)(j0 + 16 * wn + l4) * n + i0 + 16 * wm + l15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double re_q = 0.0, im_q = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double re = p1[r][c] - p2[r][c], im = (p3[r][c] - p1[r][c]) - p2[r][c];
            re += row_ror<8>(re); im += row_ror<8>(im);
            re += row_ror<4>(re); im += row_ror<4>(im);
            if (r == q) { re_q = re; im_q = im; }
        }
        cbase[(size_t)(4 * c) * n] = make_double2(re_q, im_q);
    }
}

__global__ void k_fill(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}

template<int BK> int run(const cplx* A, const cplx* B, cplx* C, int n, size_t cs, int nb, hipEvent_t ea, hipEvent_t eb, const char* what) {
    const dim3 grid((n / 64) * (n / 32) * nb), blk(512);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(ea);
        hipLaunchKernelGGL((k_gemm_kq<BK>), grid, blk, 0, 0, A, B, C, n, cs, nb);
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%s: %.1f us, %.1f TFLOP/s (8 flop per complex multiply-add)\n", what, best * 1e3, 8.0 * n * n * n * nb / best / 1e9);
    return 0;
}

int main() {
    const int n = 512, nb = 128;
    const size_t cs = (size_t)16 * 1024 * 1024 / 16;          // 16 MiB between the chains: A, B, C of 4 MiB each
    cplx* p;
    CK(hipMalloc(&p, cs * nb * 16));
    cplx *A = p, *B = p + (size_t)n * n, *C = B + (size_t)n * n;
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, cs * nb * 2, 1u);
    CK(hipDeviceSynchronize());
    std::vector<cplx> hA((size_t)n * n), hB((size_t)n * n), hC((size_t)n * n);
    CK(hipMemcpy(hA.data(), A, hA.size() * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), B, hB.size() * 16, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_gemm_kq<16>), dim3((n / 64) * (n / 32) * nb), dim3(512), 0, 0, A, B, C, n, cs, nb);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), C, hC.size() * 16, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int j = 0; j < n; j += 37)
        for (int i = 0; i < n; i += 29) {
            std::complex<double> s(0.0, 0.0);
            for (int k = 0; k < n; ++k)
                s += std::complex<double>(hA[(size_t)k * n + i].x, hA[(size_t)k * n + i].y) * std::complex<double>(hB[(size_t)j * n + k].x, hB[(size_t)j * n + k].y);
            maxerr = fmax(maxerr, std::abs(s - std::complex<double>(hC[(size_t)j * n + i].x, hC[(size_t)j * n + i].y)));
        }
    printf("k-groups GEMM: max |error| vs host product %.3e\n", maxerr);
    if (run<16>(A, B, C, n, cs, nb, ea, eb, "BK = 16")) return 1;
    if (run<32>(A, B, C, n, cs, nb, ea, eb, "BK = 32")) return 1;
    return 0;
}
