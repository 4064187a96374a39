// Developer microbenchmark (round 2): the delayed-update flush G += X Gr at the bench's shape -- 128 chains, n = 512, K per chain
// drawn like the accepted-update count of a block (2 * Binomial(32, 0.47)) -- with different ways of scheduling the tile of G.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/flush_r2.hip -o /tmp/flush_r2 && /tmp/flush_r2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void chain_tile(int tiles, int nb, int& chain, int& tile) {
    // blocks b, b+8, ... run on the same XCD: all tiles of a chain on one XCD (as xcd_chain_tile in dqmc_internal.h)
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int per = nb >> 3;                       // chains per XCD
    chain = xcd * per + slot / tiles;
    tile = slot % tiles;
}

// VAR 0: the production kernel (tile of G read 16 x 16 at a time after the MFMA loop)
// VAR 1: 3M accumulators combined first, then ALL 16 loads of the 32 x 32 tile in flight at once, then add + store
// VAR 2: VAR 1 + operand fragments requested two k-steps ahead
// VAR 3: VAR 1 with nontemporal loads / stores of G;  VAR 5: VAR 2 + nontemporal;  VAR 6: nontemporal stores only;  VAR 7: loads only
template<int VAR, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flush(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
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
            const int gi = i0 + a * 16 + l15;
            const cplx t = X[(size_t)gkc * ldx + gi];
            a_[a] = (gk < K) ? t : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int gj = j0 + b * 16 + l15;
            const cplx t = Gr[(size_t)gj * ldg + gkc];
            b_[b] = (gk < K) ? t : make_double2(0.0, 0.0);
        }
    };
    auto mac = [&](const cplx (&af)[2], const cplx (&bf)[2]) {
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
    };
    if (VAR == 2 || VAR == 5) {
        cplx a0[2], b0[2], a1[2], b1[2];
        loadab(0, a0, b0);
        loadab(4, a1, b1);
        for (int k0 = 0; k0 < K; k0 += 8) {
            cplx a2[2], b2[2], a3[2], b3[2];
            loadab(k0 + 8, a2, b2);
            mac(a0, b0);
            loadab(k0 + 12, a3, b3);
            mac(a1, b1);                       // k0 + 4 >= K: fragments are zero
#pragma unroll
            for (int a = 0; a < 2; ++a) { a0[a] = a2[a]; b0[a] = b2[a]; a1[a] = a3[a]; b1[a] = b3[a]; }
        }
    } else {
        cplx af[2], bf[2];
        loadab(0, af, bf);
        for (int k0 = 0; k0 < K; k0 += 4) {
            cplx an[2], bn[2];
            loadab(k0 + 4, an, bn);
            mac(af, bf);
#pragma unroll
            for (int a = 0; a < 2; ++a) { af[a] = an[a]; bf[a] = bn[a]; }
        }
    }
    if (VAR == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                cplx c[4];
                const int gi = i0 + a * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = j0 + b * 16 + l4 + 4 * r;
                    c[r] = G[(size_t)min(gj, n - 1) * ldc + min(gi, n - 1)];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = j0 + b * 16 + l4 + 4 * r;
                    double re = acc_re[a][b][r], im = acc_im[a][b][r];
                    { const double p1 = re, p2 = acc_p2[a][b][r]; re = p1 - p2; im = (im - p1) - p2; }
                    if (gi < n && gj < n) G[(size_t)gj * ldc + gi] = make_double2(c[r].x + re, c[r].y + im);
                }
            }
    } else {
        cplx c[2][2][4];
        cplx* base = G + (size_t)(j0 + l4) * ldc + i0 + l15;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    if (VAR == 3 || VAR == 5 || VAR == 7) { c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y); }
                    else c[a][b][r] = *p;
                }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_im[a][b] = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
                acc_re[a][b] = acc_re[a][b] - acc_p2[a][b];
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    const cplx v = make_double2(c[a][b][r].x + acc_re[a][b][r], c[a][b][r].y + acc_im[a][b][r]);
                    if (VAR == 3 || VAR == 5 || VAR == 6) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
                    else *p = v;
                }
    }
}

// VAR 4: a workgroup walks over TPW consecutive tiles of its chain; the tile of G for the NEXT tile is requested before the MFMA
// loop of the current one (64 VGPRs more: two waves per SIMD)
template<int TPW>
__global__ __launch_bounds__(256, 2) void k_flush_pipe(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                                  cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev, size_t cs, int nb) {
    const int tn = n / 64;
    int chain, grp;
    chain_tile(tn * tn / TPW, nb, chain, grp);
    X += chain * cs; Gr += chain * cs; G += chain * cs;
    int K = Kmax;
    { int kd = Kdev[chain]; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    cplx c[2][2][4];
    auto tile_base = [&](int t) {
        const int tile = grp * TPW + t;
        const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
        return G + (size_t)(j0 + l4) * ldc + i0 + l15;
    };
    auto loadc = [&](cplx* base) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) c[a][b][r] = base[(size_t)(b * 16 + 4 * r) * ldc + a * 16];
    };
    loadc(tile_base(0));
    for (int t = 0; t < TPW; ++t) {
        const int tile = grp * TPW + t;
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
                const cplx t2 = X[(size_t)gkc * ldx + i0 + a * 16 + l15];
                a_[a] = (gk < K) ? t2 : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const cplx t2 = Gr[(size_t)(j0 + b * 16 + l15) * ldg + gkc];
                b_[b] = (gk < K) ? t2 : make_double2(0.0, 0.0);
            }
        };
        cplx af[2], bf[2];
        loadab(0, af, bf);
        for (int k0 = 0; k0 < K; k0 += 4) {
            cplx an[2], bn[2];
            loadab(k0 + 4, an, bn);
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
#pragma unroll
            for (int a = 0; a < 2; ++a) { af[a] = an[a]; bf[a] = bn[a]; }
        }
        cplx* base = tile_base(t);
        cplx out[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double p1 = acc_re[a][b][r], p2 = acc_p2[a][b][r], p3 = acc_im[a][b][r];
                    out[a][b][r] = make_double2(c[a][b][r].x + (p1 - p2), c[a][b][r].y + ((p3 - p1) - p2));
                }
        if (t + 1 < TPW) loadc(tile_base(t + 1));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) base[(size_t)(b * 16 + 4 * r) * ldc + a * 16] = out[a][b][r];
    }
}

// pure read-modify-write of G with the same tiling (no product): what the memory system gives this access pattern
__global__ __launch_bounds__(256, 4) void k_rmw(cplx* __restrict__ G, int ldc, int n, size_t cs, int nb) {
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    G += chain * cs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    cplx* base = G + (size_t)(j0 + l4) * ldc + i0 + l15;
    cplx c[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = base[(size_t)((e >> 3) * 16 + 4 * (e & 3)) * ldc + ((e >> 2) & 1) * 16];
#pragma unroll
    for (int e = 0; e < 16; ++e) base[(size_t)((e >> 3) * 16 + 4 * (e & 3)) * ldc + ((e >> 2) & 1) * 16] = make_double2(c[e].x + 1.0, c[e].y - 1.0);
}

__global__ void k_fill(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-2;      // small: G stays finite over the repetitions
    }
}

struct Bufs { cplx *X, *Gr, *G; int* Kd; int n, nb; size_t cs; hipEvent_t a, b; };

template<class F> float timeit(const Bufs& B, F launch) {
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(B.a);
        launch();
        hipEventRecord(B.b); hipEventSynchronize(B.b);
        float ms; hipEventElapsedTime(&ms, B.a, B.b); if (ms < best) best = ms;
        if (rep >= 2) sum += ms;
    }
    printf("best %.1f us, mean %.1f us", best * 1e3f, sum / 6 * 1e3f);
    return best * 1e3f;
}

int main(int argc, char** argv) {
    const int n = 512, nb = 128;
    const size_t cs = (size_t)24 * 1024 * 1024 / 16;       // 24 MiB between the chains
    cplx* p; int* Kd;
    CK(hipMalloc(&p, cs * nb * 16)); CK(hipMemset(p, 0, cs * nb * 16));
    if (argc > 1 && atoi(argv[1]) == 1) {            // random operands: the matrix cores draw more power than on zeros
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, cs * nb * 2, 1u);
        CK(hipDeviceSynchronize());
        printf("random operands\n");
    }
    CK(hipMalloc(&Kd, nb * 4));
    Bufs B; B.n = n; B.nb = nb; B.cs = cs; B.G = p; B.X = p + (size_t)n * n; B.Gr = B.X + (size_t)n * 64; B.Kd = Kd;
    CK(hipEventCreate(&B.a)); CK(hipEventCreate(&B.b));
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int> hk(nb);
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
        printf("---- %s: mean K %.1f, %.0f MB RMW, %.2f GFLOP ----\n", mode == 0 ? "K ~ 2 Binomial(32, 0.47)" : "K = 64", ksum / nb, bytes / 1e6, flops / 1e9);
        const dim3 grid(64 * nb), blk(256);
        auto rep = [&](const char* name, float us) { printf("  <- %s: %.2f TB/s, %.1f TFLOP/s\n", name, bytes / us / 1e6, flops / us / 1e6); };
        rep("rmw only", timeit(B, [&] { hipLaunchKernelGGL(k_rmw, grid, blk, 0, 0, B.G, n, n, cs, nb); }));
        rep("var0 production, 2/CU hint", timeit(B, [&] { hipLaunchKernelGGL((k_flush<0, 2>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var1 batched tile, hint 2", timeit(B, [&] { hipLaunchKernelGGL((k_flush<1, 2>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var1 batched tile, hint 3", timeit(B, [&] { hipLaunchKernelGGL((k_flush<1, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var2 batched + 2-step prefetch, hint 3", timeit(B, [&] { hipLaunchKernelGGL((k_flush<2, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var3 batched nontemporal, hint 3", timeit(B, [&] { hipLaunchKernelGGL((k_flush<3, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var5 batched + 2-step prefetch + nontemporal", timeit(B, [&] { hipLaunchKernelGGL((k_flush<5, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var6 nontemporal stores only", timeit(B, [&] { hipLaunchKernelGGL((k_flush<6, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var7 nontemporal loads only", timeit(B, [&] { hipLaunchKernelGGL((k_flush<7, 3>), grid, blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
        rep("var4 pipelined 2 tiles", timeit(B, [&] { hipLaunchKernelGGL((k_flush_pipe<2>), dim3(32 * nb), blk, 0, 0, B.X, n, B.Gr, 64, B.G, n, n, 64, Kd, cs, nb); }));
    }
    return 0;
}
