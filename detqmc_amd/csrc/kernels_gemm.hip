// Complex-fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Replaces the zgemm calls behind the reference's Armadillo expressions on the stabilised sweep:
// UdV chaining U_l*U', V_t_l*V_t' (src/detmodel.h:987, :1143), the five products of greenFromUdV
// (:784-815) and the delayed-update flush g += X*Y (src/detsdwopdim.cpp:3156).
//
//   C[MxN] (+)= rowscale_i * ( op(A)[MxK] . diag(kscale^{+-1}) . op(B)[KxN] ) * colscale_j
//
// op = N or conjugate transpose.  A complex product is THREE real MFMAs per 16x16x4 step ("3M", Karatsuba):
//   P1 += ar*br,  P2 += ai*bi,  P3 += (ar + ai)*(br + bi);   re = P1 - P2,  im = P3 - P1 - P2
// instead of four (re += ar*br - ai*bi, im += ar*bi + ai*br): these kernels are bound by the fp64 matrix cores (the 4M
// form ran at 0.60-0.68 of the MFMA peak), so 25 % fewer MFMAs is 25 % less time in the MFMA loop; the two extra
// adds per fragment go to the vector ALU, which idles next to a 64-cycle MFMA.  The cancellation in `im` costs a few
// ulps relative to |a||b| -- far inside the 1e-10 parity tolerance (tests).  DQMC_GEMM_4M=1 selects the 4-MFMA form.  Operands are staged through LDS as [k][i] / [k][j]
// interleaved (re,im) so one ds_read_b128 feeds both the real and the imaginary fragment.  The MFMA
// is issued with the operand roles swapped (D^T = B^T A^T): the f64 accumulator then holds 16
// CONSECUTIVE ROWS of one column per register across lanes 0..15 -- a 256-byte contiguous run in the
// column-major C -- so the epilogue stores coalesce.
//
// Workgroup = 4 waves in a 2x2 arrangement, each wave TMxTN MFMA tiles: 32x32 block tiles when
// n_g = 512 (256 workgroups = one per CU), 64x64 for the large O(3) lattices.
#include "dqmc_internal.h"
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

template<int TM, int TN, bool M3, int TAG>
__global__ __launch_bounds__(256, 2) void k_zgemm(GemmArgs g, size_t cs, int nb) {   // 2 workgroups per CU: <= 256 registers
    constexpr int BM = 32 * TM, BN = 32 * TN, BK = 16;
    constexpr int NA = BM * BK / 256, NB_ = BN * BK / 256;      // elements of the A / B tile staged per thread
    __shared__ cplx sA[2][BK][BM + 1];                            // double buffered: the loads of tile t + 1 are in
    __shared__ cplx sB[2][BK][BN + 1];                            // flight while the MFMAs work on tile t
    const int tm = (g.M + BM - 1) / BM, tnn = (g.N + BN - 1) / BN;
    int chain, tile;
    xcd_chain_tile(tm * tnn, nb, chain, tile);
    if (!g.sharedA) g.A = chain_ptr_i(g.A, cs, chain);
    if (!g.sharedB) g.B = chain_ptr_i(g.B, cs, chain);
    g.C = chain_ptr_i(g.C, cs, chain); g.Kdev = chain_ptr_i(g.Kdev, cs, chain); g.kscale = chain_ptr_i(g.kscale, cs, chain);
    g.rowscale = chain_ptr_i(g.rowscale, cs, chain); g.colscale = chain_ptr_i(g.colscale, cs, chain);
    g.a_kgather = chain_ptr_i(g.a_kgather, cs, chain);

    int K = g.K;
    if (g.Kdev) { int kd = (*g.Kdev) * g.Kmul; K = kd < K ? kd : K; }
    if (K <= 0 && g.accumulate) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = (tile % tm) * BM, j0 = (tile / tm) * BN;
    const int l15 = lane & 15, l4 = lane >> 4;

    // 4M: acc_re / acc_im; 3M: acc_re = P1, acc_im = P3, acc_p2 = P2
    v4d acc_re[TM][TN], acc_im[TM][TN], acc_p2[M3 ? TM : 1][M3 ? TN : 1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); if (M3) acc_p2[a][b] = (v4d)(0.0); }

    cplx ra[NA], rb[NB_];
    // global -> registers: op(A) tile element (i, k), op(B) tile element (k, j).  Every load has a clamped, always valid
    // address and the bounds are applied afterwards by a select: guarded loads compile to one exec-masked branch +
    // s_waitcnt vmcnt(0) per element, which serialises the eight loads of a tile instead of keeping them in flight
    // behind the MFMAs of the previous tile.
    const int Mm1 = g.M - 1, Nm1 = g.N - 1;
    auto gload = [&](int k0) {
        const int Km1 = K - 1;
        int ka[NA];
        bool oka[NA];
        if (g.opA == 0) {
            int ia[NA], ca[NA];
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * 256;
                const int gi = i0 + idx % BM, gk = k0 + idx / BM;
                oka[e] = gi < g.M && gk < K;
                ia[e] = min(gi, Mm1); ka[e] = min(gk, Km1); ca[e] = ka[e];
            }
            if (g.a_kgather) {
#pragma unroll
                for (int e = 0; e < NA; ++e) ca[e] = g.a_kgather[ka[e]];
            }
#pragma unroll
            for (int e = 0; e < NA; ++e) ra[e] = g.A[(size_t)ca[e] * g.lda + ia[e]];
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * 256;
                const int gk = k0 + idx % BK, gi = i0 + idx / BK;
                oka[e] = gi < g.M && gk < K;
                ka[e] = min(gk, Km1);
                const cplx t = g.A[(size_t)min(gi, Mm1) * g.lda + ka[e]];
                ra[e] = make_double2(t.x, -t.y);
            }
        }
        if (g.kscale) {
            double sc[NA];
#pragma unroll
            for (int e = 0; e < NA; ++e) sc[e] = g.kscale[ka[e]];
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const double f = g.kscale_invert ? 1.0 / sc[e] : sc[e];
                ra[e].x *= f; ra[e].y *= f;
            }
        }
#pragma unroll
        for (int e = 0; e < NA; ++e) if (!oka[e]) ra[e] = make_double2(0.0, 0.0);
        if (g.opB == 0) {
#pragma unroll
            for (int e = 0; e < NB_; ++e) {
                const int idx = tid + e * 256;
                const int gk = k0 + idx % BK, gj = j0 + idx / BK;
                const cplx t = g.B[(size_t)min(gj, Nm1) * g.ldb + min(gk, Km1)];
                rb[e] = (gj < g.N && gk < K) ? t : make_double2(0.0, 0.0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < NB_; ++e) {
                const int idx = tid + e * 256;
                const int gj = j0 + idx % BN, gk = k0 + idx / BN;
                const cplx t = g.B[(size_t)min(gk, Km1) * g.ldb + min(gj, Nm1)];
                rb[e] = (gj < g.N && gk < K) ? make_double2(t.x, -t.y) : make_double2(0.0, 0.0);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            int i, k;
            if (g.opA == 0) { i = idx % BM; k = idx / BM; } else { k = idx % BK; i = idx / BK; }
            sA[buf][k][i] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < NB_; ++e) {
            const int idx = tid + e * 256;
            int j, k;
            if (g.opB == 0) { k = idx % BK; j = idx / BK; } else { j = idx % BN; k = idx / BN; }
            sB[buf][k][j] = rb[e];
        }
    };

    const int kbeg = g.b_lower ? (j0 / BK) * BK : 0;      // triangular op(B): rows above the tile's first column are zero
    if (K > kbeg) { gload(kbeg); sstore(0); }
    __syncthreads();
    for (int k0 = kbeg, buf = 0; k0 < K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < K;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            cplx af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = sA[buf][kk + l4][wm * 16 * TM + a * 16 + l15];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = sB[buf][kk + l4][wn * 16 * TN + b * 16 + l15];
            double asum[TM], bsum[TN];
            if (M3) {
#pragma unroll
                for (int a = 0; a < TM; ++a) asum[a] = af[a].x + af[a].y;
#pragma unroll
                for (int b = 0; b < TN; ++b) bsum[b] = bf[b].x + bf[b].y;
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    // roles swapped: first operand indexes the OUTPUT "row" (= column j of C)
                    if (M3) {
                        acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                        acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].y, acc_p2[a][b], 0, 0, 0);
                        acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                    } else {
                        acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                        acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bf[b].y, af[a].y, acc_re[a][b], 0, 0, 0);
                        acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].x, acc_im[a][b], 0, 0, 0);
                        acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].y, acc_im[a][b], 0, 0, 0);
                    }
                }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();       // one barrier per tile: nobody refills a buffer that a slower wave still reads
    }

    // ---- epilogue: accumulator element r of lane: out-row (=j) = l4 + 4 r, out-col (=i) = l15 ----
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int gi = i0 + wm * 16 * TM + a * 16 + l15;
            const int gic = min(gi, Mm1);
            cplx cold[4];
            if (g.accumulate) {                                    // the four loads together, bounds by select (see gload)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = j0 + wn * 16 * TN + b * 16 + l4 + 4 * r;
                    cold[r] = g.C[(size_t)min(gj, Nm1) * g.ldc + gic];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = j0 + wn * 16 * TN + b * 16 + l4 + 4 * r;
                double re = acc_re[a][b][r], im = acc_im[a][b][r];
                if (M3) { const double p1 = re, p2 = acc_p2[a][b][r]; re = p1 - p2; im = (im - p1) - p2; }
                if (g.rowscale) { double sc = g.rowscale[gic] * g.colscale[min(gj, Nm1)]; re *= sc; im *= sc; }
                else if (g.colscale) { double sc = g.colscale[min(gj, Nm1)]; re *= sc; im *= sc; }
                if (g.negate) { re = -re; im = -im; }
                if (g.accumulate) { re += cold[r].x; im += cold[r].y; }
                if (gi < g.M && gj < g.N) g.C[(size_t)gj * g.ldc + gi] = make_double2(re, im);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Delayed-update flush  G += X Gr  (reference: g += X * Y, src/detsdwopdim.cpp:3156) with K = MSF * (accepted
// updates of the block) <= 64 read on the device.  This is a read-modify-write stream over G (2 x 16 n^2 bytes)
// with a thin product riding on it, so it is built for the memory system, not for the matrix cores: no LDS, no
// barrier -- every wave pulls its MFMA operand fragments and then its 32 x 32 tile of G straight from global
// memory into registers; two workgroups per CU so that one's loads overlap the other's MFMAs.
// Fragment convention as in k_zgemm (operand roles swapped so that the stores coalesce).
// ---------------------------------------------------------------------------------------------
// FULL: n is a multiple of 32 -- every wave's 32 x 32 tile lies inside G or outside of it, no clamps or guards
template<bool M3, bool FULL>
__global__ __launch_bounds__(256, FULL ? 3 : 2) void k_flush(const cplx* __restrict__ X, int ldx, const cplx* __restrict__ Gr, int ldg,
                                                  cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev,
                                                  int Kmul, size_t cs, int nb) {
    const int tn = (n + 63) / 64;
    int chain, tile;
    xcd_chain_tile(tn * tn, nb, chain, tile);
    X = chain_ptr_i(X, cs, chain); Gr = chain_ptr_i(Gr, cs, chain); G = chain_ptr_i(G, cs, chain); Kdev = chain_ptr_i(Kdev, cs, chain);
    int K = Kmax;
    if (Kdev) { int kd = (*Kdev) * Kmul; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    if (i0 >= n || j0 >= n) return;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[M3 ? 2 : 1][M3 ? 2 : 1];      // 3M: P1, P3, P2 (see the top of this file)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); if (M3) acc_p2[a][b] = (v4d)(0.0); }
    // operand fragments of k-step k0 (clamped addresses + select: no branch, no wait per load); the fragments of step
    // k0 + 4 are requested before the MFMAs of step k0 are issued
    auto loadab = [&](int k0, cplx (&a_)[2], cplx (&b_)[2]) {
        const int gk = k0 + l4, gkc = min(gk, K - 1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int gi = i0 + a * 16 + l15;
            const cplx t = X[(size_t)gkc * ldx + (FULL ? gi : min(gi, n - 1))];
            a_[a] = (gk < K && (FULL || gi < n)) ? t : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int gj = j0 + b * 16 + l15;
            const cplx t = Gr[(size_t)(FULL ? gj : min(gj, n - 1)) * ldg + gkc];
            b_[b] = (gk < K && (FULL || gj < n)) ? t : make_double2(0.0, 0.0);
        }
    };
    auto mac = [&](const cplx (&af)[2], const cplx (&bf)[2]) {
        double asum[2], bsum[2];
        if (M3) {
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = af[a].x + af[a].y; bsum[a] = bf[a].x + bf[a].y; }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (M3) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                } else {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].x, acc_re[a][b], 0, 0, 0);
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bf[b].y, af[a].y, acc_re[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].y, af[a].x, acc_im[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b].x, af[a].y, acc_im[a][b], 0, 0, 0);
                }
            }
    };
    // two k-steps per trip, the fragments of the trip after next requested before the MFMAs of this one (a second k-step past K
    // multiplies zeros: K = MSF j is even, at most one idle step per tile)
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
    // 3M: the three accumulators are combined first (96 -> 64 registers), which makes room for the tile of G
    if (M3) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc_im[a][b] = (acc_im[a][b] - acc_re[a][b]) - acc_p2[a][b];
                acc_re[a][b] = acc_re[a][b] - acc_p2[a][b];
            }
    }
    // The tile of G is requested only now -- holding it across the MFMA loop costs 64 VGPRs, i.e. the third resident workgroup
    // per CU whose loads overlap this one's MFMAs (scripts/micro/flush_tiles.hip: 75 -> 55 us; round 2: 197 -> 212 ms per
    // 128-chain sweep when it was tried again) -- but then ALL 16 loads of the 32 x 32 tile go out together (one memory round trip
    // per wave instead of four), as nontemporal accesses: G is streamed once per launch and must not evict the X / Gr panels of its
    // chain from the L2 (scripts/micro/flush_r2.hip, 128 chains: 258 -> 218 us at K ~ 30, 357 -> 308 us at K = 64).
    if constexpr (FULL) {
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
    } else {
        // ragged edge (n not a multiple of 32): 16 x 16 at a time, clamped loads and guarded stores
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
                    if (gi < n && gj < n) G[(size_t)gj * ldc + gi] = make_double2(c[r].x + acc_re[a][b][r], c[r].y + acc_im[a][b][r]);
                }
            }
    }
}

// developer knob: DQMC_GEMM_4M=1 runs the 4-MFMA complex product (A/B measurements, rounding cross-checks)
static bool use_4m() {
    static const bool v = getenv("DQMC_GEMM_4M") && atoi(getenv("DQMC_GEMM_4M")) != 0;
    return v;
}

void launch_flush(const Launch& lc, const cplx* X, int ldx, const cplx* Gr, int ldg, cplx* G, int ldc, int n, int Kmax,
                  const int* Kdev, int Kmul) {
    const int tn = (n + 63) / 64;
    const dim3 grid = (lc.nb % 8 == 0) ? dim3(tn * tn * lc.nb, 1, 1) : dim3(tn * tn, 1, lc.nb);
    static const bool force_ragged = getenv("DQMC_FLUSH_RAGGED") && atoi(getenv("DQMC_FLUSH_RAGGED")) != 0;   // developer knob (A/B)
    const bool full = n % 32 == 0 && !force_ragged;
#define FLUSH_LAUNCH(M3_, FULL_) hipLaunchKernelGGL((k_flush<M3_, FULL_>), grid, dim3(256), 0, lc.st, X, ldx, Gr, ldg, G, ldc, n, Kmax, Kdev, Kmul, lc.cs, lc.nb)
    if (use_4m()) { if (full) FLUSH_LAUNCH(false, true); else FLUSH_LAUNCH(false, false); }
    else          { if (full) FLUSH_LAUNCH(true, true);  else FLUSH_LAUNCH(true, false); }
#undef FLUSH_LAUNCH
}

template<int TAG>
static void launch_gemm_tagged(const Launch& lc, const GemmArgs& a) {
    // fill the chip: 64x64 tiles only when they still give >= 256 workgroups
    long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * lc.nb;
    if (tiles64 >= 256 && a.N > 32 && a.M > 32) {
        const int t = ((a.M + 63) / 64) * ((a.N + 63) / 64);
        const dim3 grid = (lc.nb % 8 == 0) ? dim3(t * lc.nb, 1, 1) : dim3(t, 1, lc.nb);
        if (use_4m()) hipLaunchKernelGGL((k_zgemm<2, 2, false, TAG>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
        else          hipLaunchKernelGGL((k_zgemm<2, 2, true, TAG>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
    } else {
        const int t = ((a.M + 31) / 32) * ((a.N + 31) / 32);
        const dim3 grid = (lc.nb % 8 == 0) ? dim3(t * lc.nb, 1, 1) : dim3(t, 1, lc.nb);
        if (use_4m()) hipLaunchKernelGGL((k_zgemm<1, 1, false, TAG>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
        else          hipLaunchKernelGGL((k_zgemm<1, 1, true, TAG>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
    }
}
void launch_gemm(const Launch& lc, const GemmArgs& a) {
    if (a.tag) launch_gemm_tagged<1>(lc, a);
    else       launch_gemm_tagged<0>(lc, a);
}
