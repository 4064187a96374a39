// Developer microbenchmark (round 4): the delayed-update flush G += X GrT^T at the bench's shape -- 128 chains, n = 512, K per chain
// drawn like the accepted-update count of a block (2 * Binomial(32, 0.47)), or fixed -- production kernel (register fragments straight
// from global memory / L2, tile of G requested behind the MFMA loop) against LDS-shared operand panels:
//   var 1: the workgroup's 64 x 64 tile shares ONE copy of its operand panels X[64 rows, K8], GrT[64 rows, K8] in LDS (each L2 byte
//          fetched once per workgroup instead of twice), the tile of G is requested BEFORE the MFMA loop (fragment waits are lgkmcnt,
//          the tile's vmcnt is only waited for in the epilogue) -- 2 workgroups per CU (64 KB of LDS each, K in halves of 32)
//   var 2: var 1 with the tile requested behind the loop (separates the two effects)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/flush_r4.hip -o /tmp/flush_r4 && /tmp/flush_r4 [1 = random operands]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void chain_tile(int tiles, int nb, int& chain, int& tile) {
    const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
    chain = (t / tiles) * 8 + xcd;
    tile = t % tiles;
}

// ---- production kernel (k_flush<true, true, 1> of kernels_gemm.hip at the time of writing) ----
__global__ __launch_bounds__(256, 3) void k_flush_prod(const cplx* __restrict__ X, const cplx* __restrict__ GrT, int ld,
                                                        cplx* __restrict__ G, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    X += chain * cs; GrT += chain * cs; G += chain * cs;
    const int K = Kdev[chain];
    if (K <= 0) return;
    const int K8 = (K + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    const int ia = i0 + l15, ia2 = ia + 16, jb = j0 + l15, jb2 = jb + 16;
    const cplx* xa = X + (size_t)l4 * ld;
    const cplx* gb = GrT + (size_t)l4 * ld;
    auto loadf = [&](int k0, cplx (&f)[4]) {
        const size_t o = (size_t)k0 * ld;
        f[0] = xa[o + ia]; f[1] = xa[o + ia2]; f[2] = gb[o + jb]; f[3] = gb[o + jb2];
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    {
        cplx s[4], t[4];
        loadf(0, s);
        for (int k0 = 0; k0 < K8; k0 += 8) {
            loadf(k0 + 4, t);
            __builtin_amdgcn_sched_barrier(0);
            mac(s);
            __builtin_amdgcn_sched_barrier(0);
            loadf(min(k0 + 8, K8 - 4), s);
            __builtin_amdgcn_sched_barrier(0);
            mac(t);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc_im[a][b] = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
            acc_re[a][b] = acc_re[a][b] - acc_p2[a][b];
        }
    cplx c[2][2][4];
    cplx* base = G + (size_t)(j0 + l4) * ldc + i0 + l15;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
            }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
            }
}

// ---- LDS-shared operand panels ----
// EARLY: 1 = the tile of G is requested before the MFMA loop, 0 = behind it.  KH = k values staged per phase (32: 64 KB of LDS).
template<int EARLY, int KH, int MINB, int SKEW = 0>
__global__ __launch_bounds__(256, MINB) void k_flush_lds(const cplx* __restrict__ X, const cplx* __restrict__ GrT, int ld,
                                                          cplx* __restrict__ G, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    extern __shared__ cplx sm[];                  // Xs[KH][64], Gs[KH][64]
    cplx* Xs = sm;
    cplx* Gs = sm + KH * 64;
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    X += chain * cs; GrT += chain * cs; G += chain * cs;
    const int K = Kdev[chain];
    if (K <= 0) return;
    // SKEW: the second workgroup of every CU (dispatch order: ids 256 .. 511 of the first round) starts late, so that the two
    // workgroups of a CU run out of phase -- one in its MFMA loop while the other waits for memory
    if (SKEW > 0 && blockIdx.x >= 256 && blockIdx.x < 512) { for (int i = 0; i < SKEW; ++i) __builtin_amdgcn_s_sleep(32); }
    const int K8 = (K + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    // stage k values [kb, kb + kc) of both panels: item = (k, row), 64 consecutive rows = 1 KB contiguous in global memory and in LDS
    auto stage = [&](int kb, int kc) {
        // LDS-DMA (global_load_lds_dwordx4): one wave-instruction copies the 64 rows of one k value (1 KB contiguous in global memory)
        // to 1 KB of LDS, lane = row; no staging registers
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            __builtin_amdgcn_global_load_lds((glb_ptr)(X + k * ld + ti + lane), (lds_ptr)(Xs + kl * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(GrT + k * ld + tj + lane), (lds_ptr)(Gs + kl * 64), 16, 0, 0);
        }
    };
    auto loadf = [&](int kl, cplx (&f)[4]) {
        const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
        const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
        f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    cplx c[2][2][4];
    cplx* base = G + (size_t)(tj + wj + l4) * ldc + ti + wi + l15;
    auto load_tile = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
                }
    };
    for (int kb = 0; kb < K8; kb += KH) {
        const int kc = min(KH, K8 - kb);
        if (kb > 0) __syncthreads();              // everybody has read the previous phase's panels
        stage(kb, kc);
        __syncthreads();
        if (EARLY && kb == 0) load_tile();
        cplx s[4], t[4];
        loadf(0, s);
        for (int k0 = 0; k0 < kc; k0 += 8) {
            loadf(k0 + 4, t);
            __builtin_amdgcn_sched_barrier(0);
            mac(s);
            __builtin_amdgcn_sched_barrier(0);
            loadf(min(k0 + 8, kc - 4), s);
            __builtin_amdgcn_sched_barrier(0);
            mac(t);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc_im[a][b] = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
            acc_re[a][b] = acc_re[a][b] - acc_p2[a][b];
        }
    if (!EARLY) load_tile();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
            }
}

// var 3: LDS panels in TWO buffers of KH k values: the LDS-DMA of phase p + 1 is in flight while the MFMAs of phase p run
template<int KH, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flush_lds2(const cplx* __restrict__ X, const cplx* __restrict__ GrT, int ld,
                                                           cplx* __restrict__ G, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    X += chain * cs; GrT += chain * cs; G += chain * cs;
    const int K = Kdev[chain];
    if (K <= 0) return;
    const int K8 = (K + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int buf, int kb, int kc) {
        cplx* Xs = sm + buf * 2 * KH * 64;
        cplx* Gs = Xs + KH * 64;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            __builtin_amdgcn_global_load_lds((glb_ptr)(X + k * ld + ti + lane), (lds_ptr)(Xs + kl * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(GrT + k * ld + tj + lane), (lds_ptr)(Gs + kl * 64), 16, 0, 0);
        }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    cplx c[2][2][4];
    cplx* base = G + (size_t)(tj + wj + l4) * ldc + ti + wi + l15;
    stage(0, 0, min(KH, K8));
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
            }
    int buf = 0;
    for (int kb = 0; kb < K8; kb += KH, buf ^= 1) {
        const int kc = min(KH, K8 - kb);
        if (kb + KH < K8) stage(buf ^ 1, kb + KH, min(KH, K8 - kb - KH));
        const cplx* Xs = sm + buf * 2 * KH * 64;
        const cplx* Gs = Xs + KH * 64;
        auto loadf = [&](int kl, cplx (&f)[4]) {
            const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
            const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
            f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
        };
        cplx s[4], t[4];
        loadf(0, s);
        for (int k0 = 0; k0 < kc; k0 += 8) {
            loadf(k0 + 4, t);
            __builtin_amdgcn_sched_barrier(0);
            mac(s);
            __builtin_amdgcn_sched_barrier(0);
            loadf(min(k0 + 8, kc - 4), s);
            __builtin_amdgcn_sched_barrier(0);
            mac(t);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb + KH < K8) __syncthreads();        // next buffer landed (vmcnt(0): also the tile of G), this one free to be overwritten
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
                __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
            }
}

// LDS-DMA the compiler does not see (cdna_hip_programming.md, inline-asm recipe): lane's 16 bytes at gsrc -> LDS at (wave-uniform) dst + 16 lane
__device__ __forceinline__ void glds16_asm(const cplx* gsrc, const cplx* lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// var 5 = var 4 with the LDS-DMA hidden from the compiler (inline asm, waited for by hand): hipcc otherwise drains every outstanding
// LDS-DMA (s_waitcnt vmcnt(0)) in front of the MFMA loop's first LDS read, i.e. nothing overlaps.
// var 4 (below, kept for the record): PERSISTENT workgroups (one or two per CU) walk a list of tiles; the LDS-DMA of the NEXT 32-k chunk (the next phase of this tile or
// the first phase of the next tile) is in flight while the MFMAs of the current chunk run -- memory and matrix cores overlap inside ONE
// workgroup instead of relying on a second workgroup that tends to run in phase with the first.
template<int KH, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flush_pipe_asm(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                           cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    __shared__ int sK[128];
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;                      // chains per XCD: chain = 8 q + xcd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 256) sK[q] = Kdev[q * 8 + xcd];
    __syncthreads();
    const int T = per * tiles;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // next work item with K > 0 at or after u
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    auto dma = [&](int buf, int u, int kb) {
        const int q = u / tiles, tile = u - q * tiles;
        const size_t off = (size_t)(q * 8 + xcd) * cs;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
        cplx* Xs = sm + buf * 2 * KH * 64;
        cplx* Gs = Xs + KH * 64;
        const cplx* X = X0 + off; const cplx* GrT = GrT0 + off;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            glds16_asm(X + k * ld + ti + lane, Xs + kl * 64);
            glds16_asm(GrT + k * ld + tj + lane, Gs + kl * 64);
        }
    };
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    zero();
    cplx c[2][2][4];
    int u = skip(w), kb = 0, buf = 0;
    if (u >= T) return;
    dma(0, u, 0);
    cplx* pbase = nullptr;                        // tile whose epilogue (add + store) is still due: it runs BEHIND the next barrier, so that
                                                  // the barrier's vmcnt(0) never waits for stores that were issued a moment ago
    auto epilogue = [&]() {
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
                    cplx* p = pbase + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                    __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
                }
        zero();
    };
    while (u < T) {
        const int q = u / tiles, tile = u - q * tiles;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const bool last = kb + KH >= K8;
        // the chunk after this one
        int un = u, kbn = kb + KH;
        if (last) { un = skip(u + W); kbn = 0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this chunk's panels have landed; everybody is done with the other buffer
        // the deferred epilogue comes BEFORE the next DMA is issued: hipcc drains every outstanding LDS-DMA (vmcnt(0)) at the next use
        // of an ordinary load's result, and the epilogue uses the tile loaded one chunk ago
        if (kb == 0 && pbase) epilogue();
        if (un < T) dma(buf ^ 1, un, kbn);
        if (kb == 0) {
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + wj + l4) * ldc + (tile % tn) * 64 + wi + l15;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                        c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
                    }
            pbase = base;
        }
        {
            const cplx* Xs = sm + buf * 2 * KH * 64;
            const cplx* Gs = Xs + KH * 64;
            auto loadf = [&](int kl, cplx (&f)[4]) {
                const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
            };
            cplx s[4], t[4];
            loadf(0, s);
            for (int k0 = 0; k0 < kc; k0 += 8) {
                loadf(k0 + 4, t);
                __builtin_amdgcn_sched_barrier(0);
                mac(s);
                __builtin_amdgcn_sched_barrier(0);
                loadf(min(k0 + 8, kc - 4), s);
                __builtin_amdgcn_sched_barrier(0);
                mac(t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        u = un; kb = kbn; buf ^= 1;
    }
    epilogue();
}

template<int KH, int MINB, int NOG, int NOMFMA>
__global__ __launch_bounds__(256, MINB) void k_flush_pipe_abl(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                           cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    __shared__ int sK[128];
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;                      // chains per XCD: chain = 8 q + xcd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 256) sK[q] = Kdev[q * 8 + xcd];
    __syncthreads();
    const int T = per * tiles;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // next work item with K > 0 at or after u
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    auto dma = [&](int buf, int u, int kb) {
        const int q = u / tiles, tile = u - q * tiles;
        const size_t off = (size_t)(q * 8 + xcd) * cs;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
        cplx* Xs = sm + buf * 2 * KH * 64;
        cplx* Gs = Xs + KH * 64;
        const cplx* X = X0 + off; const cplx* GrT = GrT0 + off;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            glds16_asm(X + k * ld + ti + lane, Xs + kl * 64);
            glds16_asm(GrT + k * ld + tj + lane, Gs + kl * 64);
        }
    };
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    zero();
    cplx c[2][2][4];
    int u = skip(w), kb = 0, buf = 0;
    if (u >= T) return;
    dma(0, u, 0);
    cplx* pbase = nullptr;                        // tile whose epilogue (add + store) is still due: it runs BEHIND the next barrier, so that
                                                  // the barrier's vmcnt(0) never waits for stores that were issued a moment ago
    auto epilogue = [&]() {
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
                    cplx* p = pbase + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    if (NOG) { if (acc_re[a][b][r] == 1.2345e301) p->x = acc_im[a][b][r]; }
                    else {
                    __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                    __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y); }
                }
        zero();
    };
    while (u < T) {
        const int q = u / tiles, tile = u - q * tiles;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const bool last = kb + KH >= K8;
        // the chunk after this one
        int un = u, kbn = kb + KH;
        if (last) { un = skip(u + W); kbn = 0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this chunk's panels have landed; everybody is done with the other buffer
        // the deferred epilogue comes BEFORE the next DMA is issued: hipcc drains every outstanding LDS-DMA (vmcnt(0)) at the next use
        // of an ordinary load's result, and the epilogue uses the tile loaded one chunk ago
        if (kb == 0 && pbase) epilogue();
        if (un < T) dma(buf ^ 1, un, kbn);
        if (kb == 0) {
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + wj + l4) * ldc + (tile % tn) * 64 + wi + l15;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                        if (NOG) c[a][b][r] = make_double2(0.0, 0.0); else { c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y); }
                    }
            pbase = base;
        }
        {
            const cplx* Xs = sm + buf * 2 * KH * 64;
            const cplx* Gs = Xs + KH * 64;
            auto loadf = [&](int kl, cplx (&f)[4]) {
                const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
            };
            cplx s[4], t[4];
            loadf(0, s);
            for (int k0 = 0; k0 < (NOMFMA ? 8 : kc); k0 += 8) {
                loadf(k0 + 4, t);
                __builtin_amdgcn_sched_barrier(0);
                mac(s);
                __builtin_amdgcn_sched_barrier(0);
                loadf(min(k0 + 8, kc - 4), s);
                __builtin_amdgcn_sched_barrier(0);
                mac(t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        u = un; kb = kbn; buf ^= 1;
    }
    epilogue();
}

template<int KH, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flush_pipe_tim(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                           cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb, unsigned long long* __restrict__ dbg) {
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define TICK(i) do { unsigned long long t_ = __builtin_readcyclecounter(); tk[i] += t_ - tlast; tlast = t_; } while (0)
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    __shared__ int sK[128];
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;                      // chains per XCD: chain = 8 q + xcd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 256) sK[q] = Kdev[q * 8 + xcd];
    __syncthreads();
    const int T = per * tiles;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // next work item with K > 0 at or after u
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    auto dma = [&](int buf, int u, int kb) {
        const int q = u / tiles, tile = u - q * tiles;
        const size_t off = (size_t)(q * 8 + xcd) * cs;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
        cplx* Xs = sm + buf * 2 * KH * 64;
        cplx* Gs = Xs + KH * 64;
        const cplx* X = X0 + off; const cplx* GrT = GrT0 + off;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            glds16_asm(X + k * ld + ti + lane, Xs + kl * 64);
            glds16_asm(GrT + k * ld + tj + lane, Gs + kl * 64);
        }
    };
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    zero();
    cplx c[2][2][4];
    int u = skip(w), kb = 0, buf = 0;
    if (u >= T) return;
    dma(0, u, 0);
    cplx* pbase = nullptr;                        // tile whose epilogue (add + store) is still due: it runs BEHIND the next barrier, so that
                                                  // the barrier's vmcnt(0) never waits for stores that were issued a moment ago
    auto epilogue = [&]() {
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
                    cplx* p = pbase + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                    __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
                }
        zero();
    };
    while (u < T) {
        const int q = u / tiles, tile = u - q * tiles;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const bool last = kb + KH >= K8;
        // the chunk after this one
        int un = u, kbn = kb + KH;
        if (last) { un = skip(u + W); kbn = 0; }
        TICK(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        TICK(1);                                  // own memory operations done
        asm volatile("s_barrier" ::: "memory");
        TICK(2);                                  // everybody's
        // the deferred epilogue comes BEFORE the next DMA is issued: hipcc drains every outstanding LDS-DMA (vmcnt(0)) at the next use
        // of an ordinary load's result, and the epilogue uses the tile loaded one chunk ago
        if (kb == 0 && pbase) epilogue();
        TICK(3);
        if (un < T) dma(buf ^ 1, un, kbn);
        if (kb == 0) {
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + wj + l4) * ldc + (tile % tn) * 64 + wi + l15;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                        c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
                    }
            pbase = base;
        }
        TICK(4);
        {
            const cplx* Xs = sm + buf * 2 * KH * 64;
            const cplx* Gs = Xs + KH * 64;
            auto loadf = [&](int kl, cplx (&f)[4]) {
                const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
            };
            cplx s[4], t[4];
            loadf(0, s);
            for (int k0 = 0; k0 < kc; k0 += 8) {
                loadf(k0 + 4, t);
                __builtin_amdgcn_sched_barrier(0);
                mac(s);
                __builtin_amdgcn_sched_barrier(0);
                loadf(min(k0 + 8, kc - 4), s);
                __builtin_amdgcn_sched_barrier(0);
                mac(t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        u = un; kb = kbn; buf ^= 1;
        tk[5] += 1;
    }
    epilogue();
    TICK(0);
    if (tid == 0 && blockIdx.x < 64) for (int i = 0; i < 6; ++i) dbg[blockIdx.x * 6 + i] = tk[i];
#undef TICK
}

// var 7: WARP-SPECIALISED persistent workgroups (one per CU, 10 waves): waves 0-3 only issue LDS reads and MFMAs, waves 4-5 only the
// LDS-DMA of the operand panels, waves 6-9 only the read-modify-write of G (they take the finished 32 x 32 blocks of the compute waves
// from LDS).  No barriers: LDS counters (ds_add / ds_read) hand buffers back and forth, so a memory wave stalled in the issue of a
// global_load never holds up the MFMA stream -- the one structure var 4-6's measurements do not rule out.  Every spin is capped: a
// protocol bug ends in wrong numbers and an error word, never in a hung GPU.
// RESULT (profiles/r04_flush_micro.log): bit-identical, no time-outs, 1.6 x SLOWER than production at every K (290 vs 180 us at the bench
// mix).  Two loader waves are busy 78 % (LDS-DMA) / 89 % (register staged) of the kernel and still starve the MFMA waves 25-36 %.
__device__ __forceinline__ void ws_flag_add(unsigned* flag) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)flag, one = 1;
    asm volatile("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(a), "v"(one) : "memory");
}
__device__ __forceinline__ unsigned ws_flag_read(unsigned* flag) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)flag;
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// wait until *flag >= want (wave-uniform); returns false on time-out or when another wave has raised the abort word
__device__ __forceinline__ bool ws_wait(unsigned* flag, unsigned want, unsigned* abort_word) {
    for (int spin = 0; spin < (1 << 20); ++spin) {
        const unsigned v = __builtin_amdgcn_readfirstlane(ws_flag_read(flag));
        if ((int)(v - want) >= 0) return true;
        if ((spin & 255) == 255 && __builtin_amdgcn_readfirstlane(ws_flag_read(abort_word)) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    if ((threadIdx.x & 63) == 0) ws_flag_add(abort_word);
    return false;
}
template<int NBUF>
__global__ __launch_bounds__(640) void k_flush_ws(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                   cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb,
                                                   unsigned* __restrict__ err) {
    constexpr int KH = 16;
    extern __shared__ cplx sm[];                  // panels [NBUF][ Xs[KH][64], Gs[KH][64] ], then the hand-over buffer acc[4][32][16] (one half of every 32 x 32 block)
    __shared__ int sK[128];
    __shared__ unsigned flags[4 + 2 * 4];         // 0 acc_ready, 1 acc_done, 2 abort; 4 + b pan_ready[b], 8 + b pan_done[b]
    cplx* accb = sm + NBUF * 2 * KH * 64;
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 640) sK[q] = Kdev[q * 8 + xcd];
    if (tid < 12) flags[tid] = 0;
    __syncthreads();                              // the only barrier
    const int T = per * tiles;
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    unsigned* abortw = &flags[2];
    if (wave < 4) {
        // ------------------------------------------------ compute waves
        const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
        v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
        auto zero = [&]() {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
        };
        auto mac = [&](const cplx (&f)[4]) {
            double asum[2], bsum[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                }
        };
        zero();
        unsigned chunk = 0, half_no = 0;          // chunks and hand-over halves so far
        for (int u = skip(w); u < T; u = skip(u + W)) {
            const int K8 = (sK[u / tiles] + 7) & ~7;
            for (int kb = 0; kb < K8; kb += KH, ++chunk) {
                const int kc = min(KH, K8 - kb);
                const int buf = chunk % NBUF;
                if (!ws_wait(&flags[4 + buf], 2 * (chunk / NBUF + 1), abortw)) return;
                const cplx* Xs = sm + buf * 2 * KH * 64;
                const cplx* Gs = Xs + KH * 64;
                auto loadf = [&](int kl, cplx (&f)[4]) {
                    const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                    const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                    f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
                };
                cplx s[4], t[4];
                loadf(0, s);
                for (int k0 = 0; k0 < kc; k0 += 8) {
                    loadf(k0 + 4, t);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(s);
                    __builtin_amdgcn_sched_barrier(0);
                    loadf(min(k0 + 8, kc - 4), s);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[8 + buf]);               // this wave is done with the buffer
            }
            // hand the 32 x 32 block over in two halves (rows a 16 .. a 16 + 15): accb[wave][col][row & 15]
            cplx* ab = accb + wave * 512;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (!ws_wait(&flags[1], 4 * half_no, abortw)) return;      // the previous half has been taken by all four read-modify-write waves
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const v4d im = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
                    const v4d re = acc_re[a][b] - acc_p2[a][b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ab[(b * 16 + 4 * r + l4) * 16 + l15] = make_double2(re[r], im[r]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[0]);
                half_no += 1;
            }
            zero();
        }
    } else if (wave < 6) {
        // ------------------------------------------------ panel loaders: wave 4 the X panel, wave 5 the GrT panel.  Register staged (16 plain loads of 1 KB
        // in flight per chunk, then 16 LDS writes): a wave gets one LDS-DMA instruction out per ~ 330 cycles (phase timers of the first
        // version: the two loader waves were busy 78 % of the time issuing 2048 of them), plain loads pipeline.  Two register sets: the
        // loads of chunk q + 1 are in flight while chunk q waits for its LDS buffer and is written.
        const int which = wave - 4;
        unsigned chunk = 0;
        int u = skip(w), kb = 0;                  // the chunk whose loads are issued next
        auto chunk_src = [&](int uu, int kbb, int& kc) -> const cplx* {
            const int q = uu / tiles, tile = uu - q * tiles;
            const int K8 = (sK[q] + 7) & ~7;
            kc = min(KH, K8 - kbb);
            const size_t off = (size_t)(q * 8 + xcd) * cs;
            return (which == 0 ? X0 + off + (tile % tn) * 64 : GrT0 + off + (tile / tn) * 64) + lane + (size_t)kbb * ld;
        };
        auto advance = [&]() { const int K8 = (sK[u / tiles] + 7) & ~7; kb += KH; if (kb >= K8) { kb = 0; u = skip(u + W); } };
        double rAx[KH], rAy[KH], rBx[KH], rBy[KH];   // (arrays of double2 are not promoted to registers by this hipcc; arrays of double are)
        int kcA = 0, kcB = 0;
#define WS_ISSUE(r, kc) do { kc = 0; if (u < T) { const cplx* src_ = chunk_src(u, kb, kc); \
            _Pragma("unroll") for (int kl = 0; kl < KH; ++kl) { const cplx t_ = src_[(size_t)min(kl, kc - 1) * ld]; r##x[kl] = t_.x; r##y[kl] = t_.y; } advance(); } } while (0)
#define WS_COMMIT(r, kc) do { const int buf = chunk % NBUF; if (!ws_wait(&flags[8 + buf], 4 * (chunk / NBUF), abortw)) return; \
            cplx* dst_ = sm + buf * 2 * KH * 64 + which * KH * 64 + lane; \
            _Pragma("unroll") for (int kl = 0; kl < KH; ++kl) dst_[min(kl, kc - 1) * 64] = make_double2(r##x[kl], r##y[kl]); \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) ws_flag_add(&flags[4 + buf]); chunk += 1; } while (0)
        WS_ISSUE(rA, kcA);
        while (kcA > 0) {
            WS_ISSUE(rB, kcB);
            WS_COMMIT(rA, kcA);
            if (kcB == 0) break;
            WS_ISSUE(rA, kcA);
            WS_COMMIT(rB, kcB);
        }
#undef WS_ISSUE
#undef WS_COMMIT
    } else {
        // ------------------------------------------------ read-modify-write of G: wave 6 + m owns columns 16 m .. 16 m + 15 of the tile, lane = row
        const int mw = wave - 6;
        unsigned half_no = 0;
        for (int u = skip(w); u < T; u = skip(u + W)) {
            const int q = u / tiles, tile = u - q * tiles;
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + 16 * mw) * ldc + (tile % tn) * 64 + lane;
            cplx c[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const cplx* p = base + (size_t)j * ldc; c[j].x = __builtin_nontemporal_load(&p->x); c[j].y = __builtin_nontemporal_load(&p->y); }
            // column 16 mw + j of the tile, row = lane: compute wave (lane >> 5, (16 mw) >> 5), half (lane >> 4) & 1, row-in-half lane & 15
            const cplx* ab = accb + (((lane >> 5) * 2 + ((16 * mw) >> 5)) * 512) + ((16 * mw) & 31) * 16 + (lane & 15);
            cplx d[16];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (!ws_wait(&flags[0], 4 * (half_no + 1), abortw)) return;
                if (((lane >> 4) & 1) == a) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) d[j] = ab[j * 16];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[1]);
                half_no += 1;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                cplx* p = base + (size_t)j * ldc;
                __builtin_nontemporal_store(c[j].x + d[j].x, &p->x);
                __builtin_nontemporal_store(c[j].y + d[j].y, &p->y);
            }
        }
    }
    if (tid == 0 && flags[2] != 0) atomicAdd(err, 1u);
}

template<int NBUF>
__global__ __launch_bounds__(640) void k_flush_wst(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                   cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb,
                                                   unsigned* __restrict__ err, unsigned long long* __restrict__ dbg) {
    constexpr int KH = 16;
    unsigned long long twait = 0, tstart = __builtin_readcyclecounter(), tw0;
#define WSWAIT(f, v) (tw0 = __builtin_readcyclecounter(), wsok = ws_wait(f, v, abortw), twait += __builtin_readcyclecounter() - tw0, wsok)
    bool wsok = true;
    extern __shared__ cplx sm[];                  // panels [NBUF][ Xs[KH][64], Gs[KH][64] ], then the hand-over buffer acc[4][32][16] (one half of every 32 x 32 block)
    __shared__ int sK[128];
    __shared__ unsigned flags[4 + 2 * 4];         // 0 acc_ready, 1 acc_done, 2 abort; 4 + b pan_ready[b], 8 + b pan_done[b]
    cplx* accb = sm + NBUF * 2 * KH * 64;
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 640) sK[q] = Kdev[q * 8 + xcd];
    if (tid < 12) flags[tid] = 0;
    __syncthreads();                              // the only barrier
    const int T = per * tiles;
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    unsigned* abortw = &flags[2];
    if (wave < 4) {
        // ------------------------------------------------ compute waves
        const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
        v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
        auto zero = [&]() {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
        };
        auto mac = [&](const cplx (&f)[4]) {
            double asum[2], bsum[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                }
        };
        zero();
        unsigned chunk = 0, half_no = 0;          // chunks and hand-over halves so far
        for (int u = skip(w); u < T; u = skip(u + W)) {
            const int K8 = (sK[u / tiles] + 7) & ~7;
            for (int kb = 0; kb < K8; kb += KH, ++chunk) {
                const int kc = min(KH, K8 - kb);
                const int buf = chunk % NBUF;
                if (!WSWAIT(&flags[4 + buf], 2 * (chunk / NBUF + 1))) return;
                const cplx* Xs = sm + buf * 2 * KH * 64;
                const cplx* Gs = Xs + KH * 64;
                auto loadf = [&](int kl, cplx (&f)[4]) {
                    const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                    const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                    f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
                };
                cplx s[4], t[4];
                loadf(0, s);
                for (int k0 = 0; k0 < kc; k0 += 8) {
                    loadf(k0 + 4, t);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(s);
                    __builtin_amdgcn_sched_barrier(0);
                    loadf(min(k0 + 8, kc - 4), s);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[8 + buf]);               // this wave is done with the buffer
            }
            // hand the 32 x 32 block over in two halves (rows a 16 .. a 16 + 15): accb[wave][col][row & 15]
            cplx* ab = accb + wave * 512;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (!WSWAIT(&flags[1], 4 * half_no)) return;      // the previous half has been taken by all four read-modify-write waves
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const v4d im = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
                    const v4d re = acc_re[a][b] - acc_p2[a][b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ab[(b * 16 + 4 * r + l4) * 16 + l15] = make_double2(re[r], im[r]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[0]);
                half_no += 1;
            }
            zero();
        }
    } else if (wave < 6) {
        // ------------------------------------------------ panel loaders: wave 4 the X panel, wave 5 the GrT panel.  Register staged (16 plain loads of 1 KB
        // in flight per chunk, then 16 LDS writes): a wave gets one LDS-DMA instruction out per ~ 330 cycles (phase timers of the first
        // version: the two loader waves were busy 78 % of the time issuing 2048 of them), plain loads pipeline.  Two register sets: the
        // loads of chunk q + 1 are in flight while chunk q waits for its LDS buffer and is written.
        const int which = wave - 4;
        unsigned chunk = 0;
        int u = skip(w), kb = 0;                  // the chunk whose loads are issued next
        auto chunk_src = [&](int uu, int kbb, int& kc) -> const cplx* {
            const int q = uu / tiles, tile = uu - q * tiles;
            const int K8 = (sK[q] + 7) & ~7;
            kc = min(KH, K8 - kbb);
            const size_t off = (size_t)(q * 8 + xcd) * cs;
            return (which == 0 ? X0 + off + (tile % tn) * 64 : GrT0 + off + (tile / tn) * 64) + lane + (size_t)kbb * ld;
        };
        auto advance = [&]() { const int K8 = (sK[u / tiles] + 7) & ~7; kb += KH; if (kb >= K8) { kb = 0; u = skip(u + W); } };
        double rAx[KH], rAy[KH], rBx[KH], rBy[KH];   // (arrays of double2 are not promoted to registers by this hipcc; arrays of double are)
        int kcA = 0, kcB = 0;
#define WS_ISSUE(r, kc) do { kc = 0; if (u < T) { const cplx* src_ = chunk_src(u, kb, kc); \
            _Pragma("unroll") for (int kl = 0; kl < KH; ++kl) { const cplx t_ = src_[(size_t)min(kl, kc - 1) * ld]; r##x[kl] = t_.x; r##y[kl] = t_.y; } advance(); } } while (0)
#define WS_COMMIT(r, kc) do { const int buf = chunk % NBUF; if (!WSWAIT(&flags[8 + buf], 4 * (chunk / NBUF))) return; \
            cplx* dst_ = sm + buf * 2 * KH * 64 + which * KH * 64 + lane; \
            _Pragma("unroll") for (int kl = 0; kl < KH; ++kl) dst_[min(kl, kc - 1) * 64] = make_double2(r##x[kl], r##y[kl]); \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) ws_flag_add(&flags[4 + buf]); chunk += 1; } while (0)
        WS_ISSUE(rA, kcA);
        while (kcA > 0) {
            WS_ISSUE(rB, kcB);
            WS_COMMIT(rA, kcA);
            if (kcB == 0) break;
            WS_ISSUE(rA, kcA);
            WS_COMMIT(rB, kcB);
        }
#undef WS_ISSUE
#undef WS_COMMIT
    } else {
        // ------------------------------------------------ read-modify-write of G: wave 6 + m owns columns 16 m .. 16 m + 15 of the tile, lane = row
        const int mw = wave - 6;
        unsigned half_no = 0;
        for (int u = skip(w); u < T; u = skip(u + W)) {
            const int q = u / tiles, tile = u - q * tiles;
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + 16 * mw) * ldc + (tile % tn) * 64 + lane;
            cplx c[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const cplx* p = base + (size_t)j * ldc; c[j].x = __builtin_nontemporal_load(&p->x); c[j].y = __builtin_nontemporal_load(&p->y); }
            // column 16 mw + j of the tile, row = lane: compute wave (lane >> 5, (16 mw) >> 5), half (lane >> 4) & 1, row-in-half lane & 15
            const cplx* ab = accb + (((lane >> 5) * 2 + ((16 * mw) >> 5)) * 512) + ((16 * mw) & 31) * 16 + (lane & 15);
            cplx d[16];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (!WSWAIT(&flags[0], 4 * (half_no + 1))) return;
                if (((lane >> 4) & 1) == a) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) d[j] = ab[j * 16];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) ws_flag_add(&flags[1]);
                half_no += 1;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                cplx* p = base + (size_t)j * ldc;
                __builtin_nontemporal_store(c[j].x + d[j].x, &p->x);
                __builtin_nontemporal_store(c[j].y + d[j].y, &p->y);
            }
        }
    }
    if (tid == 0 && flags[2] != 0) atomicAdd(err, 1u);
    if (lane == 0 && blockIdx.x < 32) { dbg[(blockIdx.x * 10 + wave) * 2] = twait; dbg[(blockIdx.x * 10 + wave) * 2 + 1] = __builtin_readcyclecounter() - tstart; }
#undef WSWAIT
}

// var 6: persistent AND paced -- ONE workgroup per CU (one wave per SIMD, 128 KB of LDS: two buffers of 32 k); every vector-memory
// instruction of a tile is issued from INSIDE the MFMA loop, a few per k-step pair: the LDS-DMA lines of the next chunk, the 16 loads of
// this tile's G (needed only at the tile's end).  A wave issues in order: a burst of 16 + 16 + 16 memory instructions at a chunk boundary
// keeps it -- and with one wave per SIMD the matrix pipe -- stalled while the texture-address unit works through 48 KB per wave
// (phase timers of var 5: ~3500 cycles per chunk in "issue"); paced, each one hides behind the MFMAs already in the pipe.
// The epilogue of tile t (add + 16 stores) runs at the start of tile t + 1, behind the barrier.
template<int DUMMY>
__global__ __launch_bounds__(256, 1) void k_flush_pp(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                      cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    constexpr int KH = 32;
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    __shared__ int sK[128];
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 256) sK[q] = Kdev[q * 8 + xcd];
    __syncthreads();
    const int T = per * tiles;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    // description of the chunk whose panels are being fetched: this wave's lines are (panel, kl = wave + 4 j), j < nline
    const cplx* nX = nullptr; const cplx* nG = nullptr;      // lane's source pointers at k = kb (row ti + lane resp. tj + lane)
    int nline = 0, nbuf = 0, nissued = 0;
    auto dma_setup = [&](int buf, int u, int kb) {
        const int q = u / tiles, tile = u - q * tiles;
        const size_t off = (size_t)(q * 8 + xcd) * cs;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        nX = X0 + off + (size_t)(kb + wave) * ld + (tile % tn) * 64 + lane;
        nG = GrT0 + off + (size_t)(kb + wave) * ld + (tile / tn) * 64 + lane;
        nline = kc / 4; nbuf = buf; nissued = 0;
    };
    auto dma_some = [&](int count) {              // the next `count` k lines (both panels each)
        for (int c = 0; c < count && nissued < nline; ++c, ++nissued) {
            cplx* Xs = sm + nbuf * 2 * KH * 64 + (wave + 4 * nissued) * 64;
            glds16_asm(nX + (size_t)(4 * nissued) * ld, Xs);
            glds16_asm(nG + (size_t)(4 * nissued) * ld, Xs + KH * 64);
        }
    };
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    zero();
    cplx c[2][2][4];
    int u = skip(w), kb = 0, buf = 0;
    if (u >= T) return;
    dma_setup(0, u, 0);
    dma_some(8);
    cplx* pbase = nullptr;                        // tile whose epilogue is due
    cplx* base = nullptr;
    auto epilogue = [&]() {
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
                    cplx* p = pbase + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                    __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
                }
        zero();
    };
    // loads 4 g .. 4 g + 3 of the tile (a = g >> 1, b = g & 1: one 16 x 16 sub-tile)
#define TILE_LOADS(g) do { \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
            const cplx* p_ = base + (size_t)(((g) & 1) * 16 + 4 * r) * ldc + ((g) >> 1) * 16; \
            c[(g) >> 1][(g) & 1][r].x = __builtin_nontemporal_load(&p_->x); c[(g) >> 1][(g) & 1][r].y = __builtin_nontemporal_load(&p_->y); } } while (0)
    while (u < T) {
        const int q = u / tiles, tile = u - q * tiles;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const bool last = kb + KH >= K8;
        int un = u, kbn = kb + KH;
        if (last) { un = skip(u + W); kbn = 0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this chunk's panels have landed; everybody is done with the other buffer
        const bool first = (kb == 0);
        if (first) {
            if (pbase) epilogue();
            base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + wj + l4) * ldc + (tile % tn) * 64 + wi + l15;
            pbase = base;
        }
        const bool have_next = un < T;
        if (have_next) dma_setup(buf ^ 1, un, kbn);
        const int npairs = kc >> 3;
        const int per_pair = have_next ? (nline + npairs - 1) / npairs : 0;
        {
            const cplx* Xs = sm + buf * 2 * KH * 64;
            const cplx* Gs = Xs + KH * 64;
            auto loadf = [&](int kl, cplx (&f)[4]) {
                const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
            };
            cplx s[4], t[4];
            loadf(0, s);
            for (int k0 = 0; k0 < kc; k0 += 8) {
                loadf(k0 + 4, t);
                __builtin_amdgcn_sched_barrier(0);
                mac(s);
                __builtin_amdgcn_sched_barrier(0);
                if (per_pair) dma_some(per_pair);
                loadf(min(k0 + 8, kc - 4), s);
                __builtin_amdgcn_sched_barrier(0);
                mac(t);
                __builtin_amdgcn_sched_barrier(0);
                if (first) {
                    const int pi = k0 >> 3;
                    if (pi == 0) TILE_LOADS(0); else if (pi == 1) TILE_LOADS(1); else if (pi == 2) TILE_LOADS(2); else TILE_LOADS(3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (first) {
                if (npairs < 2) TILE_LOADS(1);
                if (npairs < 3) TILE_LOADS(2);
                if (npairs < 4) TILE_LOADS(3);
            }
        }
        u = un; kb = kbn; buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epilogue();
#undef TILE_LOADS
}

// var 4: PERSISTENT workgroups (one or two per CU) walk a list of tiles; the LDS-DMA of the NEXT 32-k chunk (the next phase of this tile or
// the first phase of the next tile) is in flight while the MFMAs of the current chunk run -- memory and matrix cores overlap inside ONE
// workgroup instead of relying on a second workgroup that tends to run in phase with the first.
template<int KH, int MINB>
__global__ __launch_bounds__(256, MINB) void k_flush_pipe(const cplx* __restrict__ X0, const cplx* __restrict__ GrT0, int ld,
                                                           cplx* __restrict__ G0, int ldc, int n, const int* __restrict__ Kdev, size_t cs, int nb) {
    extern __shared__ cplx sm[];                  // [2][ Xs[KH][64], Gs[KH][64] ]
    __shared__ int sK[128];
    const int tn = n / 64, tiles = tn * tn;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int per = nb >> 3;                      // chains per XCD: chain = 8 q + xcd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int q = tid; q < per; q += 256) sK[q] = Kdev[q * 8 + xcd];
    __syncthreads();
    const int T = per * tiles;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // next work item with K > 0 at or after u
    auto skip = [&](int u) { while (u < T && sK[u / tiles] <= 0) u += W; return u; };
    auto dma = [&](int buf, int u, int kb) {
        const int q = u / tiles, tile = u - q * tiles;
        const size_t off = (size_t)(q * 8 + xcd) * cs;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
        cplx* Xs = sm + buf * 2 * KH * 64;
        cplx* Gs = Xs + KH * 64;
        const cplx* X = X0 + off; const cplx* GrT = GrT0 + off;
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            __builtin_amdgcn_global_load_lds((glb_ptr)(X + k * ld + ti + lane), (lds_ptr)(Xs + kl * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(GrT + k * ld + tj + lane), (lds_ptr)(Gs + kl * 64), 16, 0, 0);
        }
    };
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
            }
    };
    zero();
    cplx c[2][2][4];
    int u = skip(w), kb = 0, buf = 0;
    if (u >= T) return;
    dma(0, u, 0);
    cplx* pbase = nullptr;                        // tile whose epilogue (add + store) is still due: it runs BEHIND the next barrier, so that
                                                  // the barrier's vmcnt(0) never waits for stores that were issued a moment ago
    auto epilogue = [&]() {
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
                    cplx* p = pbase + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                    __builtin_nontemporal_store(c[a][b][r].x + acc_re[a][b][r], &p->x);
                    __builtin_nontemporal_store(c[a][b][r].y + acc_im[a][b][r], &p->y);
                }
        zero();
    };
    while (u < T) {
        const int q = u / tiles, tile = u - q * tiles;
        const int K8 = (sK[q] + 7) & ~7;
        const int kc = min(KH, K8 - kb);
        const bool last = kb + KH >= K8;
        // the chunk after this one
        int un = u, kbn = kb + KH;
        if (last) { un = skip(u + W); kbn = 0; }
        __syncthreads();                          // this chunk's panels have landed (the barrier drains the LDS-DMA); everybody is done with the other buffer
        // the deferred epilogue comes BEFORE the next DMA is issued: hipcc drains every outstanding LDS-DMA (vmcnt(0)) at the next use
        // of an ordinary load's result, and the epilogue uses the tile loaded one chunk ago
        if (kb == 0 && pbase) epilogue();
        if (un < T) dma(buf ^ 1, un, kbn);
        if (kb == 0) {
            cplx* base = G0 + (size_t)(q * 8 + xcd) * cs + (size_t)((tile / tn) * 64 + wj + l4) * ldc + (tile % tn) * 64 + wi + l15;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                        c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
                    }
            pbase = base;
        }
        {
            const cplx* Xs = sm + buf * 2 * KH * 64;
            const cplx* Gs = Xs + KH * 64;
            auto loadf = [&](int kl, cplx (&f)[4]) {
                const cplx* xr = Xs + (kl + l4) * 64 + wi + l15;
                const cplx* gr = Gs + (kl + l4) * 64 + wj + l15;
                f[0] = xr[0]; f[1] = xr[16]; f[2] = gr[0]; f[3] = gr[16];
            };
            cplx s[4], t[4];
            loadf(0, s);
            for (int k0 = 0; k0 < kc; k0 += 8) {
                loadf(k0 + 4, t);
                __builtin_amdgcn_sched_barrier(0);
                mac(s);
                __builtin_amdgcn_sched_barrier(0);
                loadf(min(k0 + 8, kc - 4), s);
                __builtin_amdgcn_sched_barrier(0);
                mac(t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        u = un; kb = kbn; buf ^= 1;
    }
    epilogue();
}

__global__ void k_rmw(cplx* __restrict__ G, int ldc, int n, size_t cs, int nb) {
    const int tn = n / 64;
    int chain, tile;
    chain_tile(tn * tn, nb, chain, tile);
    G += chain * cs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    cplx* base = G + (size_t)(j0 + l4) * ldc + i0 + l15;
    cplx c[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { const cplx* p = base + (size_t)((q >> 1) * 4) * ldc + (q & 1) * 16; c[q].x = __builtin_nontemporal_load(&p->x); c[q].y = __builtin_nontemporal_load(&p->y); }
#pragma unroll
    for (int q = 0; q < 16; ++q) { cplx* p = base + (size_t)((q >> 1) * 4) * ldc + (q & 1) * 16; __builtin_nontemporal_store(c[q].x + 1.0, &p->x); __builtin_nontemporal_store(c[q].y, &p->y); }
}

__global__ void k_fill(double* p, size_t count, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
__global__ void k_diff(const cplx* a, const cplx* b, size_t count, double* out) {
    double m = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        m = fmax(m, fmax(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y)));
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(m));
}

struct Bufs { cplx *G, *X, *GrT; int* Kd; int n, nb; size_t cs; hipEvent_t a, b; };
template<class F> static float timeit(const Bufs&, hipEvent_t a, hipEvent_t b, F f, int reps = 20) {
    f(); f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const float us = ms * 1000.f / reps;
    printf("%9.1f us", us);
    return us;
}

int main(int argc, char** argv) {
    const int n = 512, nb = 128;
    const size_t cs = (size_t)24 * 1024 * 1024 / 16;       // 24 MiB between the chains
    cplx* p; int* Kd;
    CK(hipMalloc(&p, cs * nb * 16)); CK(hipMemset(p, 0, cs * nb * 16));
    cplx* G2; CK(hipMalloc(&G2, (size_t)n * n * nb * 16));
    double* dmax; CK(hipMalloc(&dmax, 8));
    const bool rnd = argc > 1 && atoi(argv[1]) == 1;
    CK(hipMalloc(&Kd, nb * 4));
    Bufs B; B.n = n; B.nb = nb; B.cs = cs; B.G = p; B.X = p + (size_t)n * n; B.GrT = B.X + (size_t)n * 64; B.Kd = Kd;
    CK(hipEventCreate(&B.a)); CK(hipEventCreate(&B.b));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<0, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<0, 16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<0, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 32, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 32, 2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 32, 2, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds<1, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds2<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_lds2<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024));
    CK(hipFuncSetAttribute((const void*)k_rmw, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_asm<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_asm<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_asm<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_tim<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pp<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_ws<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_ws<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    unsigned* wserr; CK(hipMalloc(&wserr, 4)); CK(hipMemset(wserr, 0, 4));
    CK(hipFuncSetAttribute((const void*)k_flush_wst<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    unsigned long long* wsdbg; CK(hipMalloc(&wsdbg, 32 * 10 * 2 * 8));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_abl<32, 1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_abl<32, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_abl<16, 2, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_flush_pipe_abl<16, 2, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    unsigned long long* dbg; CK(hipMalloc(&dbg, 64 * 6 * 8));
    const dim3 grid(64 * nb), blk(256);
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<int> hk(nb);
        double ksum = 0;
        srand(7);
        for (int i = 0; i < nb; ++i) {
            int acc = 0;
            for (int t = 0; t < 32; ++t) acc += (rand() % 100) < 47;
            int acc2 = acc;
            for (int t = 0; t < 32; ++t) acc2 += (rand() % 100) < 47;
            if (acc2 > 32) acc2 = 32;
            hk[i] = mode == 0 ? 2 * acc : mode == 1 ? 64 : mode == 2 ? 56 : mode == 3 ? 32 : ((i % 5) < 3 ? 2 * acc2 : 0);   // mode 4: the bench at delaySteps 32 -- K ~ 2 Bin(64, 0.47), 2 of 5 chains idle
            ksum += hk[i];
        }
        CK(hipMemcpy(Kd, hk.data(), nb * 4, hipMemcpyHostToDevice));
        const double bytes = 2.0 * 16 * n * n * nb, flops = 8.0 * n * n * ksum;
        printf("---- mode %d: mean K %.1f, %.0f MB RMW, %.2f GFLOP (8 n^2 K) ----\n", mode, ksum / nb, bytes / 1e6, flops / 1e9);
        auto rep = [&](const char* name, float us) { printf("  <- %-44s %.2f TB/s, %.1f TFLOP/s\n", name, bytes / us / 1e6, flops / us / 1e6); };
        // correctness: production vs LDS variants on random operands
        if (rnd || mode == 0) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, cs * nb * 2, 1u + mode);
            for (int c = 0; c < nb; ++c) {       // zero padding columns K .. K8-1 as the gather kernel leaves them
                const int K = hk[c], K8 = (K + 7) & ~7;
                if (K8 > K) {
                    CK(hipMemsetAsync(B.X + c * cs + (size_t)K * n, 0, (size_t)(K8 - K) * n * 16, 0));
                    CK(hipMemsetAsync(B.GrT + c * cs + (size_t)K * n, 0, (size_t)(K8 - K) * n * 16, 0));
                }
            }
            std::vector<cplx> ref((size_t)n * n), got((size_t)n * n);
            auto snapshot = [&](cplx* dst) { for (int c = 0; c < nb; ++c) (void)hipMemcpyAsync(dst + (size_t)c * n * n, B.G + c * cs, (size_t)n * n * 16, hipMemcpyDeviceToDevice, 0); };
            cplx* G0; CK(hipMalloc(&G0, (size_t)n * n * nb * 16));
            snapshot(G0);
            auto restore = [&]() { for (int c = 0; c < nb; ++c) (void)hipMemcpyAsync(B.G + c * cs, G0 + (size_t)c * n * n, (size_t)n * n * 16, hipMemcpyDeviceToDevice, 0); };
            hipLaunchKernelGGL(k_flush_prod, grid, blk, 0, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb);
            snapshot(G2);
            auto check = [&](const char* nm) {
                cplx* G3; (void)hipMalloc(&G3, (size_t)n * n * nb * 16);
                snapshot(G3);
                (void)hipMemset(dmax, 0, 8);
                hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, 0, G2, G3, (size_t)n * n * nb, dmax);
                double d; (void)hipMemcpy(&d, dmax, 8, hipMemcpyDeviceToHost);
                printf("  max |%s - production| = %.3e\n", nm, d);
                (void)hipFree(G3);
            };
            restore(); hipLaunchKernelGGL((k_flush_lds<1, 32, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("lds early KH32");
            restore(); hipLaunchKernelGGL((k_flush_lds<0, 32, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("lds late KH32");
            restore(); hipLaunchKernelGGL((k_flush_lds<1, 64, 1>), grid, blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("lds early KH64");
            restore(); hipLaunchKernelGGL((k_flush_lds2<16, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("lds2 KH16");
            restore(); hipLaunchKernelGGL((k_flush_pipe<32, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("pipe KH32 256 WG");
            restore(); hipLaunchKernelGGL((k_flush_pipe_asm<32, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("pipe asm KH32 256 WG");
            restore(); hipLaunchKernelGGL((k_flush_pp<0>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("persistent paced");
            restore(); hipLaunchKernelGGL((k_flush_ws<3>), dim3(256), dim3(640), 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb, wserr); check("warp-specialised 3 buffers");
            { unsigned e = 0; CK(hipMemcpy(&e, wserr, 4, hipMemcpyDeviceToHost)); printf("  warp-specialised: %u workgroups timed out\n", e); }
            restore(); hipLaunchKernelGGL((k_flush_pipe_asm<16, 2>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("pipe asm KH16 512 WG");
            restore(); hipLaunchKernelGGL((k_flush_pipe<16, 2>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("pipe KH16 512 WG");
            restore(); hipLaunchKernelGGL((k_flush_lds2<8, 2>), grid, blk, 32 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); check("lds2 KH8");
            CK(hipDeviceSynchronize());
            CK(hipFree(G0));
            if (!rnd) CK(hipMemset(p, 0, cs * nb * 16));
        }
        rep("rmw only", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL(k_rmw, grid, blk, 0, 0, B.G, n, n, cs, nb); }));
        rep("rmw only, 2 WG/CU (64 KB LDS each)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL(k_rmw, grid, blk, 64 * 1024, 0, B.G, n, n, cs, nb); }));
        rep("rmw only, 4 WG/CU (32 KB LDS each)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL(k_rmw, grid, blk, 32 * 1024, 0, B.G, n, n, cs, nb); }));
        rep("production (register fragments, 3 WG/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL(k_flush_prod, grid, blk, 0, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 32, tile early, 2 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 32, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 32, early, skew 3 (~2.5 us)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 32, 2, 3>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 32, early, skew 6 (~5 us)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 32, 2, 6>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 32, early, skew 10 (~8 us)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 32, 2, 10>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 16, tile late, 3 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<0, 16, 3>), grid, blk, 32 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 8, tile late, 3 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<0, 8, 3>), grid, blk, 16 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 32, tile late, 2 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<0, 32, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 16, tile early, 2 WG/CU hint", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 16, 2>), grid, blk, 32 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS 2 buffers KH 16, tile early, 2 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds2<16, 2>), grid, blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS 2 buffers KH 8, tile early, 2 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds2<8, 2>), grid, blk, 32 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("WARP-SPECIALISED 3 buffers x 16 k, 256 WG x 10 waves", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_ws<3>), dim3(256), dim3(640), 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb, wserr); }));
        {
            hipLaunchKernelGGL((k_flush_wst<3>), dim3(256), dim3(640), 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb, wserr, wsdbg);
            unsigned long long h[32 * 10 * 2]; CK(hipMemcpy(h, wsdbg, sizeof(h), hipMemcpyDeviceToHost));
            double wsum[10] = {0}, tsum[10] = {0};
            for (int g = 0; g < 32; ++g) for (int wv = 0; wv < 10; ++wv) { wsum[wv] += (double)h[(g * 10 + wv) * 2] / 32; tsum[wv] += (double)h[(g * 10 + wv) * 2 + 1] / 32; }
            printf("  warp-specialised, cycles per wave (mean of 32 workgroups) waiting for a flag / total:");
            for (int wv = 0; wv < 10; ++wv) printf(" [%d] %.0fk/%.0fk", wv, wsum[wv] / 1e3, tsum[wv] / 1e3);
            printf("\n");
        }
        rep("WARP-SPECIALISED 2 buffers x 16 k, 256 WG x 10 waves", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_ws<2>), dim3(256), dim3(640), 96 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb, wserr); }));
        rep("persistent PACED KH 32, 256 WG (1/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pp<0>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent asm-DMA KH 32, 256 WG (1/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_asm<32, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent asm-DMA KH 16, 512 WG (2/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_asm<16, 2>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent asm-DMA KH 16, 256 WG (1/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_asm<16, 1>), dim3(256), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        {
            hipLaunchKernelGGL((k_flush_pipe_tim<32, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb, dbg);
            unsigned long long h[64 * 6]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            double sum[6] = {0, 0, 0, 0, 0, 0};
            for (int wgi = 0; wgi < 64; ++wgi) for (int i = 0; i < 6; ++i) sum[i] += (double)h[wgi * 6 + i] / 64;
            printf("  persistent asm-DMA KH 32, cycles (100 MHz counter) per workgroup, mean of 64: mfma+tail %.0f, own vm wait %.0f, barrier %.0f, epilogue %.0f, dma issue + tile loads %.0f; chunks %.1f\n",
                   sum[0], sum[1], sum[2], sum[3], sum[4], sum[5]);
        }
        rep("  ablation KH 32 1/CU: no G traffic", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_abl<32, 1, 1, 0>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("  ablation KH 32 1/CU: 1/4.. MFMA (8 k per chunk max)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_abl<32, 1, 0, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("  ablation KH 16 2/CU: no G traffic", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_abl<16, 2, 1, 0>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("  ablation KH 16 2/CU: MFMA 8 k per chunk max", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe_abl<16, 2, 0, 1>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent pipeline KH 32, 256 WG (1/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe<32, 1>), dim3(256), blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent pipeline KH 16, 512 WG (2/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe<16, 2>), dim3(512), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("persistent pipeline KH 16, 256 WG (1/CU)", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_pipe<16, 1>), dim3(256), blk, 64 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
        rep("LDS panels KH 64, tile early, 1 WG/CU", timeit(B, B.a, B.b, [&] { hipLaunchKernelGGL((k_flush_lds<1, 64, 1>), grid, blk, 128 * 1024, 0, B.X, B.GrT, n, B.G, n, n, Kd, cs, nb); }));
    }
    return 0;
}
