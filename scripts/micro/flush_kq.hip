// Developer microbenchmark (round 2): flush G += X Gr on v_mfma_f64_4x4x4_4b_f64 with the four blocks of the instruction used as four
// K-GROUPS of the same 4 x 4 block of G (no operand preparation on the vector ALU: every operand register comes straight from LDS and
// is used by four MFMAs); the four partial sums per element live in different lanes and are added once per tile.
// Measured on this box (scripts/micro/mfma_peak.hip): 4x4x4_4b sustains 73 TFLOP/s, 16x16x4 47.5 -- when nothing else is issued.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/flush_kq.hip -o /tmp/flush_kq && /tmp/flush_kq
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

// workgroup = 4 waves = a 32 x 32 tile of G, wave (wm, wn) its 16 x 16 quarter; K in chunks of 16 through LDS (double buffered)
#define BK 16
#define PITCH 33
template<int WPE>
__global__ __launch_bounds__(256, WPE) void k_flush_kq(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                                     cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev, size_t cs, int nb) {
    __shared__ cplx sX[2][BK][PITCH];       // [k][row]
    __shared__ cplx sG[2][BK][PITCH];       // [k][col]
    const int tn = n / 32;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    X += chain * cs; Gr += chain * cs; G += chain * cs;
    int K = Kmax;
    { int kd = Kdev[chain]; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = (tile % tn) * 32, j0 = (tile / tn) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int jj = lane & 3, q = (lane >> 2) & 3;

    // the wave's 16 x 16 tile of G, requested first (4 elements per lane): column j0 + 16 wn + 4 c + l4, row i0 + 16 wm + l15
    cplx* gbase = G + (size_t)(j0 + 16 * wn + l4) * ldc + i0 + 16 * wm + l15;
    cplx gt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const cplx* p = gbase + (size_t)(4 * c) * ldc; gt[c].x = __builtin_nontemporal_load(&p->x); gt[c].y = __builtin_nontemporal_load(&p->y); }

    double p1[4][4], p2[4][4], p3[4][4];     // [r][c]: partial sums (k-group q of this lane) of G[4 r + jj][4 c + l4] of the wave's tile
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { p1[r][c] = 0.0; p2[r][c] = 0.0; p3[r][c] = 0.0; }

    // staging: chunk k0: X[k0 + k][i0 + row] for k < 16, row < 32 (2 per thread), Gr[j0 + col][k0 + k] (2 per thread)
    const int xr = tid & 31, xk = tid >> 5;            // + 8 for the second element
    const int gk = tid & 15, gc = tid >> 4;            // + 16 columns for the second element
    cplx st[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + xk + 8 * e;
            const cplx t = X[(size_t)min(k, K - 1) * ldx + i0 + xr];
            st[e] = k < K ? t : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + gk;
            const cplx t = Gr[(size_t)(j0 + gc + 16 * e) * ldg + min(k, K - 1)];
            st[2 + e] = k < K ? t : make_double2(0.0, 0.0);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 2; ++e) sX[buf][xk + 8 * e][xr] = st[e];
#pragma unroll
        for (int e = 0; e < 2; ++e) sG[buf][gk][gc + 16 * e] = st[2 + e];
    };
    gload(0); sstore(0);
    __syncthreads();
    for (int k0 = 0, buf = 0; k0 < K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < K;
        if (more) gload(k0 + BK);
        // operand registers: k = 4 q + l4 of the chunk; first operand (index -> lane / 16 of D): Gr columns 4 c + jj;
        // second operand (index -> lane % 4 of D): X rows 4 r + jj
        cplx xf[4], gf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xf[r] = sX[buf][4 * q + l4][16 * wm + 4 * r + jj];
#pragma unroll
        for (int c = 0; c < 4; ++c) gf[c] = sG[buf][4 * q + l4][16 * wn + 4 * c + jj];
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
        if (more) sstore(buf ^ 1);
        __syncthreads();
    }
    // 3M combine, then the sum over the four k-groups (lanes 4 and 8 apart within a row of 16), then lane (jj, q, l4) keeps r = q:
    // D lane (jj, q, i = l4) of accumulator [r][c] is element (row 4 r + jj, column 4 c + l4) -- and lane (l15 = 4 q + jj, l4)
    // stores row l15 of column 4 c + l4, i.e. r = q
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
        cplx* p = gbase + (size_t)(4 * c) * ldc;
        __builtin_nontemporal_store(gt[c].x + re_q, &p->x);
        __builtin_nontemporal_store(gt[c].y + im_q, &p->y);
    }
}

__global__ void k_fill(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}

int main() {
    const int n = 512, nb = 128;
    const size_t cs = (size_t)24 * 1024 * 1024 / 16;
    cplx* p; int* Kd;
    CK(hipMalloc(&p, cs * nb * 16));
    CK(hipMalloc(&Kd, nb * 4));
    cplx *G = p, *X = p + (size_t)n * n, *Gr = X + (size_t)n * 64;
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    std::vector<int> hk(nb, 64);
    hk[0] = 38;                                              // a ragged K for the checked chain
    CK(hipMemcpy(Kd, hk.data(), nb * 4, hipMemcpyHostToDevice));
    const dim3 grid(256 * nb), blk(256);
    std::vector<cplx> hG((size_t)n * n), hX((size_t)n * 64), hGr((size_t)64 * n), out((size_t)n * n);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, cs * nb * 2, 1u);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hG.data(), G, hG.size() * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hX.data(), X, hX.size() * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hGr.data(), Gr, hGr.size() * 16, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_flush_kq<2>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), G, out.size() * 16, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int j = 0; j < n; j += 5)
        for (int i = 0; i < n; i += 3) {
            std::complex<double> s(hG[(size_t)j * n + i].x, hG[(size_t)j * n + i].y);
            for (int k = 0; k < hk[0]; ++k)
                s += std::complex<double>(hX[(size_t)k * n + i].x, hX[(size_t)k * n + i].y) * std::complex<double>(hGr[(size_t)j * 64 + k].x, hGr[(size_t)j * 64 + k].y);
            maxerr = fmax(maxerr, std::abs(s - std::complex<double>(out[(size_t)j * n + i].x, out[(size_t)j * n + i].y)));
        }
    printf("4x4x4_4b, k-groups: max |error| vs host product %.3e (K = %d)\n", maxerr, hk[0]);
    for (int mode = 0; mode < 2; ++mode) {
        double ksum = 0;
        srand(7);
        for (int i = 0; i < nb; ++i) {
            int acc = 0;
            for (int t = 0; t < 32; ++t) acc += (rand() % 100) < 47;
            hk[i] = mode == 0 ? 2 * acc : 64;
            ksum += hk[i];
        }
        CK(hipMemcpy(Kd, hk.data(), nb * 4, hipMemcpyHostToDevice));
        const double bytes = 2.0 * 16 * n * n * nb, flops = 8.0 * n * n * ksum;
        printf("---- mean K %.1f ----\n", ksum / nb);
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, (size_t)n * n * 2, 1u);
                hipEventRecord(ea);
                if (variant == 0) hipLaunchKernelGGL((k_flush_kq<2>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
                else              hipLaunchKernelGGL((k_flush_kq<3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
                hipEventRecord(eb); hipEventSynchronize(eb);
                float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
            }
            printf("k-groups, %d waves/SIMD allowed: %.1f us, %.2f TB/s, %.1f TFLOP/s (8 flop per complex multiply-add)\n", variant == 0 ? 2 : 3, best * 1e3, bytes / best / 1e9, flops / best / 1e9);
        }
    }
    return 0;
}
