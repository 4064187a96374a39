// Developer microbenchmark (round 2): the flush G += X Gr with the complex 3M product issued as v_mfma_f64_4x4x4_4b_f64
// (73 TFLOP/s sustained on this box, scripts/micro/mfma_peak.hip) instead of v_mfma_f64_16x16x4_f64 (47.5): one 16x16x4 product =
// four 4x4x4_4b products whose second operand is rotated by 0 / 4 / 8 / 12 lanes within each row of 16 lanes (DPP row_ror).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/flush_4x4.hip -o /tmp/flush_4x4 && /tmp/flush_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <complex>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void chain_tile(int tiles, int nb, int& chain, int& tile) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    chain = (slot / tiles) * 8 + xcd;
    tile = slot % tiles;
}
// value of lane (lane + 4 S) % 16 of the same row of 16 lanes (or the other direction -- the epilogue formula is fitted to it)
template<int S> __device__ __forceinline__ double row_rot(double v) {
    if (S == 0) return v;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + 4 * S, 0xf, 0xf, true);      // every lane is written: no "old" value to set up
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + 4 * S, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
struct Rot4 { double v[4]; };
__device__ __forceinline__ Rot4 rot4(double x) { Rot4 r; r.v[0] = x; r.v[1] = row_rot<1>(x); r.v[2] = row_rot<2>(x); r.v[3] = row_rot<3>(x); return r; }

// MODE 0: 16x16x4 (as production);  MODE 1: 4x4x4_4b with rotated second operand
template<int MODE, int SIGN, int WPE>
__global__ __launch_bounds__(256, WPE) void k_flush(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                                  cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev, size_t cs, int nb) {
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    X += chain * cs; Gr += chain * cs; G += chain * cs;
    int K = Kmax;
    { int kd = Kdev[chain]; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    auto loadab = [&](int k0, cplx (&a_)[2], cplx (&b_)[2]) {
        const int gk = k0 + l4, gkc = min(gk, K - 1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const cplx t = X[(size_t)gkc * ldx + i0 + a * 16 + l15];
            a_[a] = (gk < K) ? t : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const cplx t = Gr[(size_t)(j0 + b * 16 + l15) * ldg + gkc];
            b_[b] = (gk < K) ? t : make_double2(0.0, 0.0);
        }
    };
    auto mac = [&](const cplx (&af)[2], const cplx (&bf)[2]) {
        if (MODE == 0) {
            double asum[2], bsum[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = af[a].x + af[a].y; bsum[a] = bf[a].x + bf[a].y; }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                }
        } else {
            double bsum[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) bsum[b] = bf[b].x + bf[b].y;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const Rot4 ax = rot4(af[a].x), ay = rot4(af[a].y), as = rot4(af[a].x + af[a].y);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc_re[a][b][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(bf[b].x, ax.v[s], acc_re[a][b][s], 0, 0, 0);
                        acc_p2[a][b][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(bf[b].y, ay.v[s], acc_p2[a][b][s], 0, 0, 0);
                        acc_im[a][b][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(bsum[b], as.v[s], acc_im[a][b][s], 0, 0, 0);
                    }
            }
        }
    };
    cplx a0[2], b0[2], a1[2], b1[2];
    loadab(0, a0, b0);
    loadab(4, a1, b1);
    for (int k0 = 0; k0 < K; k0 += 8) {
        cplx a2[2], b2[2], a3[2], b3[2];
        loadab(k0 + 8, a2, b2);
        mac(a0, b0);
        loadab(k0 + 12, a3, b3);
        if (k0 + 4 < K) mac(a1, b1);
#pragma unroll
        for (int a = 0; a < 2; ++a) { a0[a] = a2[a]; b0[a] = b2[a]; a1[a] = a3[a]; b1[a] = b3[a]; }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc_im[a][b] = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
            acc_re[a][b] = acc_re[a][b] - acc_p2[a][b];
        }
    // element s of the accumulators of sub-tile (a, b):
    //   MODE 0: row = l15, col = l4 + 4 s
    //   MODE 1: first operand (Gr, column of G) index 4 q + i with q = (lane / 4) % 4, i = lane / 16; second operand (X, row of G)
    //           index 4 ((q + SIGN s) % 4) + lane % 4
    const int q = (lane >> 2) & 3, jj = lane & 3;
    cplx c[2][2][4];
    auto ptr = [&](int a, int b, int s) -> cplx* {
        int row, col;
        if (MODE == 0) { row = l15; col = l4 + 4 * s; }
        else { row = 4 * ((q + SIGN * s + 4) & 3) + jj; col = 4 * q + l4; }
        return G + (size_t)(j0 + b * 16 + col) * ldc + i0 + a * 16 + row;
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const cplx* p = ptr(a, b, s);
                c[a][b][s].x = __builtin_nontemporal_load(&p->x); c[a][b][s].y = __builtin_nontemporal_load(&p->y);
            }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                cplx* p = ptr(a, b, s);
                __builtin_nontemporal_store(c[a][b][s].x + acc_re[a][b][s], &p->x);
                __builtin_nontemporal_store(c[a][b][s].y + acc_im[a][b][s], &p->y);
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
    CK(hipMemcpy(Kd, hk.data(), nb * 4, hipMemcpyHostToDevice));
    const dim3 grid(64 * nb), blk(256);
    // ---- correctness: chain 0, against the host product ----
    std::vector<cplx> hG((size_t)n * n), hX((size_t)n * 64), hGr((size_t)64 * n), out((size_t)n * n);
    for (int variant = 0; variant < 3; ++variant) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, cs * nb * 2, 1u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hG.data(), G, hG.size() * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hX.data(), X, hX.size() * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hGr.data(), Gr, hGr.size() * 16, hipMemcpyDeviceToHost));
        if (variant == 0) hipLaunchKernelGGL((k_flush<0, 1, 3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
        if (variant == 1) hipLaunchKernelGGL((k_flush<1, 1, 3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
        if (variant == 2) hipLaunchKernelGGL((k_flush<1, -1, 3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), G, out.size() * 16, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int j = 0; j < n; j += 7)
            for (int i = 0; i < n; i += 3) {
                std::complex<double> s(hG[(size_t)j * n + i].x, hG[(size_t)j * n + i].y);
                for (int k = 0; k < 64; ++k)
                    s += std::complex<double>(hX[(size_t)k * n + i].x, hX[(size_t)k * n + i].y) * std::complex<double>(hGr[(size_t)j * 64 + k].x, hGr[(size_t)j * 64 + k].y);
                maxerr = fmax(maxerr, std::abs(s - std::complex<double>(out[(size_t)j * n + i].x, out[(size_t)j * n + i].y)));
            }
        printf("variant %d (%s): max |error| vs host product %.3e\n", variant, variant == 0 ? "16x16x4" : variant == 1 ? "4x4x4_4b, sign +" : "4x4x4_4b, sign -", maxerr);
    }
    // ---- timing ----
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
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, (size_t)n * n * 2, 1u);    // keep G of chain 0 finite
                hipEventRecord(ea);
                if (variant == 0) hipLaunchKernelGGL((k_flush<0, 1, 3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
                else if (variant == 1) hipLaunchKernelGGL((k_flush<1, -1, 3>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
                else              hipLaunchKernelGGL((k_flush<1, -1, 2>), grid, blk, 0, 0, X, n, Gr, 64, G, n, n, 64, Kd, cs, nb);
                hipEventRecord(eb); hipEventSynchronize(eb);
                float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
            }
            printf("%s: %.1f us, %.2f TB/s, %.1f TFLOP/s (8 flop per complex multiply-add)\n", variant == 0 ? "16x16x4 " : variant == 1 ? "4x4x4_4b (3 waves/SIMD)" : "4x4x4_4b (2 waves/SIMD)", best * 1e3, bytes / best / 1e9, flops / best / 1e9);
        }
    }
    return 0;
}
