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
#include <mutex>
#include <algorithm>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

// OPA / OPB (op = conjugate transpose) are template parameters: as run-time flags the two addressing variants of each operand were
// two branches in the tile loop, and the compiler -- reusing the registers of one variant as addresses of the other -- waited for
// ALL outstanding loads (s_waitcnt vmcnt(0)) between them, in front of the MFMA block of every tile.
template<int TM, int TN, bool M3, int TAG, bool OPA, bool OPB>
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
    // split-K: this workgroup contracts k in [kslice0, K) only (K shortened to the end of its slice); raw sums go to g.part
    int kslice0 = 0;
    if (g.ksplit > 1) {
        const int kc = ((K + g.ksplit - 1) / g.ksplit + 15) & ~15;
        kslice0 = blockIdx.y * kc;
        K = min(K, kslice0 + kc);
        g.part = chain_ptr_i(g.part, cs, chain);
    }

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
    double rsc[NA];
    bool oka[NA], okb[NB_];
    // global -> registers: op(A) tile element (i, k), op(B) tile element (k, j).  Every load has a clamped, always valid
    // address; NOTHING is done with the loaded values here -- bounds (select), conjugation and the k-scale are applied in
    // sstore(), i.e. AFTER the MFMAs of the tile before.  (Round 2 applied the select right after the load: the compiler put
    // s_waitcnt vmcnt(..) + v_cndmask in front of the MFMA block, so every tile waited for its successor's loads before it
    // started computing -- the double buffering overlapped nothing.)
    const int Mm1 = g.M - 1, Nm1 = g.N - 1;
    // column gather of A (a_kgather): the indices of a tile are requested ONE TILE AHEAD, so the dependent index -> element
    // chain never sits in front of an MFMA block
    // Optional operands (gather list, k-scale) are loaded UNCONDITIONALLY, from a stand-in address inside A when absent: a
    // conditional load leaves a phi (loaded value | old value) behind, and the compiler guards the "old value" arm with
    // s_waitcnt vmcnt(0) -- in the common case without k-scale that wait sat right behind the tile's A loads.
    const bool has_gather = !OPA && g.a_kgather != nullptr, has_kscale = g.kscale != nullptr;
    const int* kgather = has_gather ? g.a_kgather : (const int*)g.A;          // K ints resp. K doubles are inside A (>= K columns of 16 lda bytes)
    const double* kscale = has_kscale ? g.kscale : (const double*)g.A;
    int cnext[NA];
    auto gidx = [&](int k0) {
        if (!OPA) {
#pragma unroll
            for (int e = 0; e < NA; ++e) cnext[e] = kgather[min(k0 + (tid + e * 256) / BM, K - 1)];
        }
    };
    auto gload = [&](int k0) {
        const int Km1 = K - 1;
        int ka[NA];
        if (!OPA) {
            int ia[NA], ca[NA];
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * 256;
                const int gi = i0 + idx % BM, gk = k0 + idx / BM;
                oka[e] = gi < g.M && gk < K;
                ia[e] = min(gi, Mm1); ka[e] = min(gk, Km1); ca[e] = has_gather ? cnext[e] : ka[e];
            }
#pragma unroll
            for (int e = 0; e < NA; ++e) ra[e] = g.A[(size_t)ca[e] * g.lda + ia[e]];
            gidx(k0 + BK);
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * 256;
                const int gk = k0 + idx % BK, gi = i0 + idx / BK;
                oka[e] = gi < g.M && gk < K;
                ka[e] = min(gk, Km1);
                ra[e] = g.A[(size_t)min(gi, Mm1) * g.lda + ka[e]];
            }
        }
#pragma unroll
        for (int e = 0; e < NA; ++e) rsc[e] = kscale[ka[e]];
        if (!OPB) {
#pragma unroll
            for (int e = 0; e < NB_; ++e) {
                const int idx = tid + e * 256;
                const int gk = k0 + idx % BK, gj = j0 + idx / BK;
                rb[e] = g.B[(size_t)min(gj, Nm1) * g.ldb + min(gk, Km1)];
                okb[e] = gj < g.N && gk < K;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NB_; ++e) {
                const int idx = tid + e * 256;
                const int gj = j0 + idx % BN, gk = k0 + idx / BN;
                rb[e] = g.B[(size_t)min(gk, Km1) * g.ldb + min(gj, Nm1)];
                okb[e] = gj < g.N && gk < K;
            }
        }
    };
    auto sstore = [&](int buf) {
        constexpr double csgnA = OPA ? -1.0 : 1.0, csgnB = OPB ? -1.0 : 1.0;       // op = conjugate transpose
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            int i, k;
            if (!OPA) { i = idx % BM; k = idx / BM; } else { k = idx % BK; i = idx / BK; }
            cplx v = make_double2(ra[e].x, csgnA * ra[e].y);
            if (has_kscale) { const double f = g.kscale_invert ? 1.0 / rsc[e] : rsc[e]; v.x *= f; v.y *= f; }
            if (!oka[e]) v = make_double2(0.0, 0.0);
            sA[buf][k][i] = v;
        }
#pragma unroll
        for (int e = 0; e < NB_; ++e) {
            const int idx = tid + e * 256;
            int j, k;
            if (!OPB) { k = idx % BK; j = idx / BK; } else { j = idx % BN; k = idx / BN; }
            sB[buf][k][j] = okb[e] ? make_double2(rb[e].x, csgnB * rb[e].y) : make_double2(0.0, 0.0);
        }
    };

    const int kbeg = g.ksplit > 1 ? kslice0 : (g.b_lower ? (j0 / BK) * BK : 0);      // triangular op(B): rows above the tile's first column are zero
    if (K > kbeg) { gidx(kbeg); gload(kbeg); sstore(0); }
    __syncthreads();
    for (int k0 = kbeg, buf = 0; k0 < K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < K;
        if (more) gload(k0 + BK);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay in front of the MFMA block, their first use behind it
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
        __builtin_amdgcn_sched_barrier(0);
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
                if (g.ksplit > 1) {
                    if (gi < g.M && gj < g.N) g.part[((size_t)blockIdx.y * g.N + gj) * g.M + gi] = make_double2(re, im);
                    continue;
                }
                if (g.rowscale) { double sc = g.rowscale[gic] * g.colscale[min(gj, Nm1)]; re *= sc; im *= sc; }
                else if (g.colscale) { double sc = g.colscale[min(gj, Nm1)]; re *= sc; im *= sc; }
                if (g.negate) { re = -re; im = -im; }
                if (g.accumulate) { re += cold[r].x; im += cold[r].y; }
                if (gi < g.M && gj < g.N) g.C[(size_t)gj * g.ldc + gi] = make_double2(re, im);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Delayed-update flush  G += X GrT^T  (reference: g += X * Y, src/detsdwopdim.cpp:3156) with K = MSF * (accepted
// updates of the block) <= 64 read on the device.  A read-modify-write stream over G (2 x 16 n^2 bytes) with a thin
// product riding on it: no LDS, no barrier -- every wave pulls its MFMA operand fragments and then its 32 x 32 tile of G
// straight from global memory into registers.  Fragment convention as in k_zgemm (operand roles swapped so that the
// stores coalesce).  Both operands are n x K8 column-major (k_update_gather writes Gr transposed), K8 = K rounded up to a
// multiple of 8 with ZERO padding, so the k loop has no bounds check at all and every fragment load is a 256-byte run.
//
// Round 3: the k loop is branch free.  Round 2's version guarded its fragment loads with `k < K ? load : 0`; the compiler
// turned each guard into an exec-masked branch around the load and, unable to count outstanding loads across the branches,
// put `s_waitcnt vmcnt(0)` in front of the MFMAs of EVERY trip (ISA: fl.s:374) -- the fragments requested "two steps ahead"
// were waited for at once, and the only overlap of memory and matrix cores came from the other resident waves.
// ---------------------------------------------------------------------------------------------
// FULL: n is a multiple of 32 -- every wave's 32 x 32 tile lies inside G or outside of it, no clamps or guards
// TAG only gives the launches of the LU factorisation (trailing updates, K = 32) their own kernel name in profiles
template<bool M3, bool FULL, int STAGE, int TAG>
__global__ __launch_bounds__(256, (FULL && STAGE == 1) ? 3 : 2) void k_flush(const cplx* __restrict__ X, const cplx* __restrict__ GrT, int ld,
                                                  cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev,
                                                  int Kmul, size_t cs, int nb) {
    const int tn = (n + 63) / 64;
    int chain, tile;
    xcd_chain_tile(tn * tn, nb, chain, tile);
    X = chain_ptr_i(X, cs, chain); GrT = chain_ptr_i(GrT, cs, chain); G = chain_ptr_i(G, cs, chain); Kdev = chain_ptr_i(Kdev, cs, chain);
    int K = Kmax;
    if (Kdev) { int kd = (*Kdev) * Kmul; K = kd < K ? kd : K; }
    if (K <= 0) return;
    const int K8 = (K + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (tile % tn) * 64 + (wave >> 1) * 32, j0 = (tile / tn) * 64 + (wave & 1) * 32;
    if (i0 >= n || j0 >= n) return;
    v4d acc_re[2][2], acc_im[2][2], acc_p2[M3 ? 2 : 1][M3 ? 2 : 1];      // 3M: P1, P3, P2 (see the top of this file)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); if (M3) acc_p2[a][b] = (v4d)(0.0); }
    // fragments of one k-step: f[0], f[1] = X[i0 + {0, 16} + l15, k0 + l4];  f[2], f[3] = GrT[j0 + {0, 16} + l15, k0 + l4].
    // Rows past n (ragged tiles only) are clamped: they feed accumulator entries that are never stored.
    const int ia = FULL ? i0 + l15 : min(i0 + l15, n - 1), ia2 = FULL ? ia + 16 : min(i0 + 16 + l15, n - 1);
    const int jb = FULL ? j0 + l15 : min(j0 + l15, n - 1), jb2 = FULL ? jb + 16 : min(j0 + 16 + l15, n - 1);
    const cplx* xa = X + (size_t)l4 * ld;
    const cplx* gb = GrT + (size_t)l4 * ld;
    auto loadf = [&](int k0, cplx (&f)[4]) {
        const size_t o = (size_t)k0 * ld;
        f[0] = xa[o + ia]; f[1] = xa[o + ia2]; f[2] = gb[o + jb]; f[3] = gb[o + jb2];
    };
    auto mac = [&](const cplx (&f)[4]) {
        double asum[2], bsum[2];
        if (M3) {
#pragma unroll
            for (int a = 0; a < 2; ++a) { asum[a] = f[a].x + f[a].y; bsum[a] = f[2 + a].x + f[2 + a].y; }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (M3) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_p2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].y, acc_p2[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bsum[b], asum[a], acc_im[a][b], 0, 0, 0);
                } else {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].x, acc_re[a][b], 0, 0, 0);
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f[2 + b].y, f[a].y, acc_re[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].y, f[a].x, acc_im[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[2 + b].x, f[a].y, acc_im[a][b], 0, 0, 0);
                }
            }
    };
    // Software pipeline: two fragment sets that swap roles, written out twice per loop iteration so that no register copy closes
    // the cycle -- with `cur = next` at the end of a trip the register coalescer merges the two sets and the loads land right in
    // front of the MFMAs that use them.  The scheduling barriers keep each batch of loads in front of the OTHER set's MFMAs.  A
    // step past the end requests the last fragments again (uniform clamp of the index instead of a branch around the loads).
    //   STAGE 1: a set = one k-step (4 k, 12 MFMAs between request and use); 150 VGPRs, three workgroups per CU
    //   STAGE 2: a set = two k-steps (24 MFMAs between request and use); 188 VGPRs, two workgroups per CU
    if constexpr (STAGE == 1) {
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
    } else {
        cplx s0[4], s1[4], t0[4], t1[4];
        loadf(0, s0);
        loadf(4, s1);
        for (int k0 = 0; k0 < K8; k0 += 16) {
            const int kn = min(k0 + 8, K8 - 8);
            loadf(kn, t0);
            loadf(kn + 4, t1);
            __builtin_amdgcn_sched_barrier(0);
            mac(s0);
            mac(s1);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 8 >= K8) break;
            const int kn2 = min(k0 + 16, K8 - 8);
            loadf(kn2, s0);
            loadf(kn2 + 4, s1);
            __builtin_amdgcn_sched_barrier(0);
            mac(t0);
            mac(t1);
            __builtin_amdgcn_sched_barrier(0);
        }
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
    // per CU whose loads overlap this one's MFMAs -- but then ALL 16 loads of the 32 x 32 tile go out together (one memory round
    // trip per wave instead of four), as nontemporal accesses: G is streamed once per launch and must not evict the X / GrT
    // panels of its chain from the L2.
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

// ---------------------------------------------------------------------------------------------
// Round 4: the flush with LDS-shared operand panels (n a multiple of 32).  The four waves of a workgroup (64 x 64 tile of G) used to
// pull the SAME operand rows from the L2 twice each (8.9 TB/s of L2 -> CU traffic per launch against 4.4 TB/s of G traffic) and, having
// to wait for fragment loads with the in-order vmcnt counter in every trip, could not have their tile of G in flight behind the MFMAs.
// Now: the workgroup's panels X[64 rows, <= 32 k] and GrT[64 rows, <= 32 k] are copied ONCE into LDS by LDS-DMA (global_load_lds_dwordx4:
// one wave-instruction = the 64 rows of one k = 1 KB contiguous on both sides, no staging registers), the tile of G is requested right
// behind the first barrier -- the fragment reads of the MFMA loop are LDS reads (lgkmcnt), so the tile's loads (vmcnt) stay in flight
// until the epilogue -- and K > 32 runs in phases of 32 through the same 64 KB (two workgroups per CU).  Same MFMAs in the same order as
// k_flush: bit-identical G.  scripts/micro/flush_r4.hip, profiles/r04_flush_micro.log: 128 chains, n = 512, K ~ 2 Binomial(32, 0.47):
// 214 -> 198 us per launch (5.0 -> 5.4 TB/s of read-modify-write traffic); K = 32: 216 -> 187 us; K = 56 / 64: 272 / 297 us, unchanged.
// Used for K <= 32 (launch_flush).  What the same harness measured and dropped on the way to "MFMA time and HBM time overlap" (DESIGN.md
// section 15 has the account): the tile requested behind the loop (no gain over k_flush: what pays is the overlap, not the halved L2
// traffic); two LDS buffers of 16 or 8 k with the next phase's DMA in flight (216 / 232 us: the extra barriers cost more than the exposed
// DMA of a phase); all of K in LDS at one workgroup per CU (227 us); PERSISTENT workgroups (one wave per SIMD, the next chunk's DMA in
// flight behind the MFMAs; 308 us at K = 56 against 272) -- their phase timers show the waves stalled 3500 cycles per chunk ISSUING
// memory instructions (a wave issues in order: while the memory pipe is backed up it cannot issue MFMAs either), pacing those
// instructions through the MFMA loop made it worse (379 us), so on this chip the overlap has to come from OTHER waves of the same SIMD,
// i.e. from occupancy, which is what k_flush's three workgroups per CU already buy; a start skew between the two workgroups of a CU
// (-4 %).
// ---------------------------------------------------------------------------------------------
#define FLUSH_KH 32
template<int TAG>
__global__ __launch_bounds__(256, 2) void k_flush_lds(const cplx* __restrict__ X, const cplx* __restrict__ GrT, int ld,
                                                       cplx* __restrict__ G, int ldc, int n, int Kmax, const int* __restrict__ Kdev,
                                                       int Kmul, size_t cs, int nb) {
    extern __shared__ cplx flush_sm[];            // Xs[FLUSH_KH][64], Gs[FLUSH_KH][64]
    cplx* Xs = flush_sm;
    cplx* Gs = flush_sm + FLUSH_KH * 64;
    const int tn = (n + 63) / 64;
    int chain, tile;
    xcd_chain_tile(tn * tn, nb, chain, tile);
    X = chain_ptr_i(X, cs, chain); GrT = chain_ptr_i(GrT, cs, chain); G = chain_ptr_i(G, cs, chain); Kdev = chain_ptr_i(Kdev, cs, chain);
    int K = Kmax;
    if (Kdev) { int kd = (*Kdev) * Kmul; K = kd < K ? kd : K; }
    if (K <= 0) return;                                               // uniform over the workgroup
    const int K8 = (K + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ti = (tile % tn) * 64, tj = (tile / tn) * 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    // n = 32 (mod 64): the last tile row / column holds 32 valid rows -- the waves beyond them stage and meet the barriers, nothing else
    const bool active = (ti + wi < n) && (tj + wj < n);
    v4d acc_re[2][2], acc_im[2][2], acc_p2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc_re[a][b] = (v4d)(0.0); acc_im[a][b] = (v4d)(0.0); acc_p2[a][b] = (v4d)(0.0); }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int ri = min(ti + lane, n - 1), rj = min(tj + lane, n - 1);     // rows past n: a valid address, the values are never used
    auto stage = [&](int kb, int kc) {
        for (int kl = wave; kl < kc; kl += 4) {
            const size_t k = (size_t)(kb + kl);
            __builtin_amdgcn_global_load_lds((glb_ptr)(X + k * ld + ri), (lds_ptr)(Xs + kl * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(GrT + k * ld + rj), (lds_ptr)(Gs + kl * 64), 16, 0, 0);
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
    for (int kb = 0; kb < K8; kb += FLUSH_KH) {
        const int kc = min(FLUSH_KH, K8 - kb);
        if (kb > 0) __syncthreads();              // everybody has read the previous phase's panels
        stage(kb, kc);
        __syncthreads();                          // the panels have landed (the barrier drains the LDS-DMA)
        if (!active) continue;
        if (kb == 0) {
            // the tile of G: all 16 loads of the 32 x 32 sub-tile go out together, nontemporal (G is streamed once per launch and must
            // not evict the operand panels of its chain from the L2); they are waited for in the epilogue only
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx* p = base + (size_t)(b * 16 + 4 * r) * ldc + a * 16;
                        c[a][b][r].x = __builtin_nontemporal_load(&p->x); c[a][b][r].y = __builtin_nontemporal_load(&p->y);
                    }
        }
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
    if (!active) return;
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

// developer knob: DQMC_GEMM_4M=1 runs the 4-MFMA complex product (A/B measurements, rounding cross-checks)
static bool use_4m() {
    static const bool v = dev_knob("DQMC_GEMM_4M") && atoi(dev_knob("DQMC_GEMM_4M")) != 0;
    return v;
}

static void launch_flush_impl(const Launch& lc, const cplx* X, const cplx* GrT, int ld, cplx* G, int ldc, int n, int Kmax,
                              const int* Kdev, int Kmul, int tag);
void launch_flush(const Launch& lc, const cplx* X, const cplx* GrT, int ld, cplx* G, int ldc, int n, int Kmax,
                  const int* Kdev, int Kmul, int tag) {
    const bool sub = tag && lc.sub;
    if (sub) lc.sub->begin(lc.sub->user, SUBFAM_LU_UPDATE);
    launch_flush_impl(lc, X, GrT, ld, G, ldc, n, Kmax, Kdev, Kmul, tag);
    if (sub) lc.sub->end(lc.sub->user, SUBFAM_LU_UPDATE, 8.0 * n * n * Kmax * lc.nb, 16.0 * (2.0 * n * n + 2.0 * n * Kmax) * lc.nb);
}
static void launch_flush_impl(const Launch& lc, const cplx* X, const cplx* GrT, int ld, cplx* G, int ldc, int n, int Kmax,
                              const int* Kdev, int Kmul, int tag) {
    const int tn = (n + 63) / 64;
    const dim3 grid = (lc.nb % 8 == 0) ? dim3(tn * tn * lc.nb, 1, 1) : dim3(tn * tn, 1, lc.nb);
    static const bool force_ragged = dev_knob("DQMC_FLUSH_RAGGED") && atoi(dev_knob("DQMC_FLUSH_RAGGED")) != 0;   // developer knob (A/B)
    const bool full = n % 32 == 0 && !force_ragged;
    static const int stage = dev_knob("DQMC_FLUSH_STAGE") ? atoi(dev_knob("DQMC_FLUSH_STAGE")) : 1;                 // developer knob (A/B)
#define FLUSH_LAUNCH(M3_, FULL_, ST_) do { if (tag) hipLaunchKernelGGL((k_flush<M3_, FULL_, ST_, 1>), grid, dim3(256), 0, lc.st, X, GrT, ld, G, ldc, n, Kmax, Kdev, Kmul, lc.cs, lc.nb); \
                                           else hipLaunchKernelGGL((k_flush<M3_, FULL_, ST_, 0>), grid, dim3(256), 0, lc.st, X, GrT, ld, G, ldc, n, Kmax, Kdev, Kmul, lc.cs, lc.nb); } while (0)
    static const bool flush_4m = dev_knob("DQMC_FLUSH_4M") && atoi(dev_knob("DQMC_FLUSH_4M")) != 0;               // developer knob (A/B): 4 MFMAs, 122 registers, 4 workgroups per CU
    static const bool flush_reg = dev_knob("DQMC_FLUSH_LDS") && atoi(dev_knob("DQMC_FLUSH_LDS")) == 0;            // developer knob (A/B): the register-fragment kernel of round 3
    // K <= 32 (delay depth <= 16 for O(1) / O(2), <= 8 for O(3); the LU's trailing updates): one LDS phase, 187 against 216 us per launch of
    // 128 chains at K = 32.  Deeper blocks (the bench runs K ~ 60 of at most 64): both kernels tie within 3 % (profiles/r04_flush_micro.log,
    // modes 2 and 4), the register kernel with its three workgroups per CU stays
    static const bool flush_lds_all = dev_knob("DQMC_FLUSH_LDS") && atoi(dev_knob("DQMC_FLUSH_LDS")) == 2;      // developer knob (A/B): LDS panels for every K
    if (full && !use_4m() && !flush_4m && stage != 2 && !flush_reg && (Kmax <= FLUSH_KH || flush_lds_all)) {
        // LDS-shared operand panels; 64 KB of dynamic LDS: the attribute belongs to (function, device)
        const size_t lds = (size_t)2 * FLUSH_KH * 64 * sizeof(cplx);
        static std::mutex mu;
        static bool raised[64][2] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(mu);
            bool& r = raised[dev & 63][tag ? 1 : 0];
            if (!r) {
                const void* f = tag ? (const void*)k_flush_lds<1> : (const void*)k_flush_lds<0>;
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) r = true;
                else (void)hipGetLastError();      // the launch below then reports the problem
            }
        }
        if (tag) hipLaunchKernelGGL((k_flush_lds<1>), grid, dim3(256), lds, lc.st, X, GrT, ld, G, ldc, n, Kmax, Kdev, Kmul, lc.cs, lc.nb);
        else     hipLaunchKernelGGL((k_flush_lds<0>), grid, dim3(256), lds, lc.st, X, GrT, ld, G, ldc, n, Kmax, Kdev, Kmul, lc.cs, lc.nb);
        return;
    }
    if (use_4m() || flush_4m) { if (full) FLUSH_LAUNCH(false, true, 1); else FLUSH_LAUNCH(false, false, 1); }
    else if (stage == 2) { if (full) FLUSH_LAUNCH(true, true, 2);  else FLUSH_LAUNCH(true, false, 2); }
    else          { if (full) FLUSH_LAUNCH(true, true, 1);  else FLUSH_LAUNCH(true, false, 1); }
#undef FLUSH_LAUNCH
}

// C (+)= (-) sum over the slices of part[slice][N][M], slices added in their fixed order (deterministic)
__global__ void k_gemm_reduce(const cplx* __restrict__ part, int ksplit, int M, int N, cplx* __restrict__ C, int ldc, int accumulate, int negate, size_t cs) {
    CHAIN(part); CHAIN(C);
    const size_t total = (size_t)M * N;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % M), j = (int)(idx / M);
        double re = 0.0, im = 0.0;
        for (int sl = 0; sl < ksplit; ++sl) { const cplx v = part[(size_t)sl * total + idx]; re += v.x; im += v.y; }
        if (negate) { re = -re; im = -im; }
        cplx* c = C + (size_t)j * ldc + i;
        if (accumulate) { re += c->x; im += c->y; }
        *c = make_double2(re, im);
    }
}

template<int TM, int TN, bool M3, int TAG>
static void launch_gemm_ops(const Launch& lc, const GemmArgs& a, dim3 grid) {
    if (!a.opA) {
        if (!a.opB) hipLaunchKernelGGL((k_zgemm<TM, TN, M3, TAG, false, false>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
        else        hipLaunchKernelGGL((k_zgemm<TM, TN, M3, TAG, false, true>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
    } else {
        if (!a.opB) hipLaunchKernelGGL((k_zgemm<TM, TN, M3, TAG, true, false>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
        else        hipLaunchKernelGGL((k_zgemm<TM, TN, M3, TAG, true, true>), grid, dim3(256), 0, lc.st, a, lc.cs, lc.nb);
    }
}
template<int TAG>
static void launch_gemm_tagged(const Launch& lc, const GemmArgs& a) {
    // fill the chip: 64x64 tiles only when they still give >= 256 workgroups
    long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * lc.nb;
    if (tiles64 >= 256 && a.N > 32 && a.M > 32) {
        const int t = ((a.M + 63) / 64) * ((a.N + 63) / 64);
        const dim3 grid = (lc.nb % 8 == 0) ? dim3(t * lc.nb, 1, 1) : dim3(t, 1, lc.nb);
        if (use_4m() && TAG == 0) launch_gemm_ops<2, 2, false, 0>(lc, a, grid);
        else                      launch_gemm_ops<2, 2, true, TAG>(lc, a, grid);
    } else {
        const int t = ((a.M + 31) / 32) * ((a.N + 31) / 32);
        dim3 grid = (lc.nb % 8 == 0) ? dim3(t * lc.nb, 1, 1) : dim3(t, 1, lc.nb);
        // split-K: few tiles, long contraction, a scratch buffer offered and no epilogue scaling -> slices of >= 64 k until ~ 512
        // workgroups are in flight (a workgroup's k loop is a chain of dependent trips to memory, 16 k per trip: a slice of 288 k took
        // 30 us whatever the tile count; round 3 split into at most 8 slices of >= 256 k and only below 128 tiles)
        int ks = 1;
        const long wg = (long)t * lc.nb;
        if (a.part && a.K >= 512 && !a.Kdev && !a.kscale && !a.rowscale && !a.colscale && !a.b_lower && wg < 256) {
            const long cap = (long)((a.part_count ? a.part_count : (size_t)a.M * a.N * 8) / ((size_t)a.M * a.N));
            ks = (int)std::min<long>(std::min<long>(std::min<long>(32, cap), a.K / 64), (512 + wg - 1) / wg);
            if (ks < 2) ks = 1;
        }
        if (ks > 1) {
            GemmArgs g2 = a;
            g2.ksplit = ks; g2.accumulate = 0; g2.negate = 0;
            grid.y = ks;
            launch_gemm_ops<1, 1, true, TAG>(lc, g2, grid);
            hipLaunchKernelGGL(k_gemm_reduce, dim3(std::min(256, (a.M * a.N + 255) / 256), 1, lc.nb), dim3(256), 0, lc.st, a.part, ks, a.M, a.N,
                               a.C, a.ldc, a.accumulate, a.negate, lc.cs);
            return;
        }
        if (use_4m() && TAG == 0) launch_gemm_ops<1, 1, false, 0>(lc, a, grid);
        else                      launch_gemm_ops<1, 1, true, TAG>(lc, a, grid);
    }
}
void launch_gemm(const Launch& lc, const GemmArgs& a) {
    if (a.tag) {
        if (lc.sub) lc.sub->begin(lc.sub->user, SUBFAM_FACT_GEMM);
        launch_gemm_tagged<1>(lc, a);
        if (lc.sub) lc.sub->end(lc.sub->user, SUBFAM_FACT_GEMM, 8.0 * a.M * a.N * a.K * lc.nb,
                                16.0 * ((double)a.M * a.K + (double)a.K * a.N + (a.accumulate ? 2.0 : 1.0) * a.M * a.N) * lc.nb);
    } else launch_gemm_tagged<0>(lc, a);
}
