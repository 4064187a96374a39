// Householder QR on gfx950 (complex fp64) -- the factorisation behind the "qr" stabilisation mode.
//
// The reference stabilises B-matrix chains with udvDecompose = full SVD (src/udv.h:68-102); its own
// header keeps a QR variant as dead code (src/udv.h:150-159) and BASELINE.json's north star asks for a
// QR-based UDV.  In "qr" mode a chain matrix M (columns graded by the previous scales) is decomposed as
//        M P = Q R,   d = |diag R|,   T = D^-1 R P^T        =>   M = Q D T
// with P a column pre-pivoting (columns sorted by decreasing norm, decided on the device), Q unitary,
// and T well conditioned.  G is decomposition independent, so it agrees with the SVD path / the reference
// to rounding (tests: 1e-10), at a fraction of the cost of a Jacobi SVD.
//
// Blocked right-looking algorithm, panel width NB = 16:
//   k_qr_panel    one workgroup factors a (rows x 16) panel held in REGISTERS (each thread owns RPT rows
//                 of all 16 columns): per column one norm reduction and one batched reduction of the
//                 v^H a_c' / v_k^H v_c dot products (DPP wave sums + one LDS hop), then the rank-1 update of
//                 the rest of the panel; also builds the compact-WY factor T (zlarft) on the fly.
//   trailing update and formation of Q: three small GEMMs per panel on the MFMA kernel
//                 (W = V^H C, W <- -T^(H) W, C += V W).
#include "dqmc_internal.h"
#include <cstring>

#define QR_NB 16

template<int CTRL, int ROWMASK>
__device__ __forceinline__ double q_dpp_add(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double q_readlane_d(double x, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double q_wave_total(double v) {
    v = q_dpp_add<0xB1, 0xf>(v);
    v = q_dpp_add<0x4E, 0xf>(v);
    v = q_dpp_add<0x114, 0xf>(v);
    v = q_dpp_add<0x118, 0xf>(v);
    v = q_dpp_add<0x142, 0xa>(v);
    v = q_dpp_add<0x143, 0xc>(v);
    return q_readlane_d(v, 63);
}

// Sum 32 doubles per lane across the wavefront with a value-splitting butterfly: at every level a lane keeps
// one half of its values and trades the other half with its partner, so the traffic is 16+8+4+2+1+1 = 32
// exchanged doubles instead of 32 x 6.  Afterwards lane l (and l ^ 32) holds the total of the value with index
// bitreverse5(l & 31) in val[0].
template<int H, int MASK>
__device__ __forceinline__ void q_butterfly_level(double (&val)[32], int lane) {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        double keep = hi ? val[j + H] : val[j];
        double send = hi ? val[j] : val[j + H];
        val[j] = keep + __shfl_xor(send, MASK, 64);
    }
}
__device__ __forceinline__ void q_wave_sum32(double (&val)[32], int lane) {
    q_butterfly_level<16, 1>(val, lane);
    q_butterfly_level<8, 2>(val, lane);
    q_butterfly_level<4, 4>(val, lane);
    q_butterfly_level<2, 8>(val, lane);
    q_butterfly_level<1, 16>(val, lane);
    val[0] += __shfl_xor(val[0], 32, 64);
}
__device__ __forceinline__ cplx q_cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// One column step of the panel factorisation (C = column index inside the panel, compile time).
template<int C>
struct QrPanelStep {
    template<class S> __device__ static __forceinline__ void run(S& s) {
        s.template column<C>();
        QrPanelStep<C + 1>::run(s);
    }
};
template<>
struct QrPanelStep<QR_NB> {
    template<class S> __device__ static __forceinline__ void run(S&) {}
};

// sum of one scratch entry over the NW waves of the workgroup (pairwise, the same order for every thread)
template<int NW>
__device__ __forceinline__ double q_xwave(const double (*r)[2 * QR_NB], int idx) {
    double s = (r[0][idx] + r[1][idx]) + (r[2][idx] + r[3][idx]);
#pragma unroll
    for (int w = 4; w < NW; w += 4) s += (r[w][idx] + r[w + 1][idx]) + (r[w + 2][idx] + r[w + 3][idx]);
    return s;
}

// NT threads per workgroup: 256 (rows <= 1024) or 512 (rows <= 2560) with the panel in registers, 1024 for rows <= 4096
// (128 registers per lane there: the compiler spills part of the panel to scratch -- matrices that large spend their
// time in the trailing update, not here)
template<int RPT, int NT>
struct QrPanelState {
    static constexpr int NW = NT / 64;
    cplx a[RPT][QR_NB];          // rows tid + r*NT (relative to the panel's first row) of the panel
    double (*red)[NW][2 * QR_NB]; // [parity][wave][...] cross-wave scratch
    cplx* sT;                    // [NB][NB] compact-WY factor, column major in LDS
    cplx* sTau;                  // [NB]
    double* sBeta;               // [NB]  diagonal of R
    cplx* sAlpha;                // broadcast slot for the pivot element
    int tid, lane, wave, rows, ncols;

    template<int C>
    __device__ __forceinline__ void column() {
        if (C >= ncols) return;                     // short last panel (n not a multiple of NB)
        // ---- pivot element and norm of the part below it ----
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            int row = tid + r * NT;
            if (row > C && row < rows) part += a[r][C].x * a[r][C].x + a[r][C].y * a[r][C].y;
        }
        part = q_wave_total(part);
        if (tid == C) *sAlpha = a[0][C];            // row C lives in thread C, r = 0 (NB <= 256)
        if (lane == 0) red[0][wave][0] = part;
        __syncthreads();
        const double xnorm2 = q_xwave<NW>(red[0], 0);
        const cplx alpha = *sAlpha;
        // zlarfg: beta = -sign(Re alpha) sqrt(|alpha|^2 + xnorm^2), tau = (beta - alpha)/beta, v = x/(alpha - beta)
        cplx tau = make_double2(0.0, 0.0), scal = make_double2(0.0, 0.0);
        double beta = alpha.x;
        const bool trivial = (xnorm2 == 0.0 && alpha.y == 0.0);
        if (!trivial) {
            double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2);
            beta = (alpha.x >= 0.0) ? -nrm : nrm;
            tau = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
            double dr = alpha.x - beta, di = alpha.y, dn = dr * dr + di * di;
            scal = make_double2(dr / dn, -di / dn);                 // 1 / (alpha - beta)
        }
        // ---- v in place (zeros above row C, one at row C) ----
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            int row = tid + r * NT;
            if (row > C) a[r][C] = q_cmul(a[r][C], scal);
        }
        // the thread owning row C keeps R[C][C] = beta aside; its v entry is 1
        cplx vpiv = make_double2(1.0, 0.0);
        // ---- batched dot products: w[c'] = v^H a_c' (c' > C) and z[k] = v_k^H v_C (k < C) ----
        double val[32];                     // val[2c] = Re, val[2c+1] = Im of the partial sum for column c
#pragma unroll
        for (int c = 0; c < 32; ++c) val[c] = 0.0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            int row = tid + r * NT;
            if (row >= C && row < rows) {
                cplx v = (row == C) ? vpiv : a[r][C];
#pragma unroll
                for (int c = 0; c < QR_NB; ++c) {
                    if (c == C) continue;
                    // columns c < C hold earlier reflectors v_c below their diagonal (rows > c); row == c is 1,
                    // rows < c are R entries and do not belong to v_c.  Since row >= C > c here, a[r][c] is v_c.
                    cplx x = a[r][c];
                    if (c > C) {            // conj(v) * a
                        val[2 * c] += v.x * x.x + v.y * x.y;
                        val[2 * c + 1] += v.x * x.y - v.y * x.x;
                    } else {                // conj(v_c) * v
                        val[2 * c] += x.x * v.x + x.y * v.y;
                        val[2 * c + 1] += x.x * v.y - x.y * v.x;
                    }
                }
            }
        }
        q_wave_sum32(val, lane);
        if (lane < 32) red[1][wave][__brev((unsigned)lane) >> 27] = val[0];
        __syncthreads();
        cplx w[QR_NB];
#pragma unroll
        for (int c = 0; c < QR_NB; ++c) {
            if (c == C) { w[c] = make_double2(0.0, 0.0); continue; }
            w[c] = make_double2(q_xwave<NW>(red[1], 2 * c), q_xwave<NW>(red[1], 2 * c + 1));
        }
        // ---- apply H^H = I - conj(tau) v v^H to the rest of the panel ----
        const cplx ctau = make_double2(tau.x, -tau.y);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            int row = tid + r * NT;
            if (row >= C && row < rows) {
                cplx v = (row == C) ? vpiv : a[r][C];
                cplx tv = q_cmul(ctau, v);
#pragma unroll
                for (int c = C + 1; c < QR_NB; ++c) {
                    cplx t = q_cmul(tv, w[c]);
                    a[r][c].x -= t.x;
                    a[r][c].y -= t.y;
                }
            }
        }
        // ---- compact WY factor (zlarft, forward/columnwise): T[0:C,C] = -tau T[0:C,0:C] z, T[C,C] = tau ----
        if (tid < QR_NB) {
            int i = tid;
            cplx t = make_double2(0.0, 0.0);
            if (i < C) {
                cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < QR_NB; ++k) {
                    if (k < C && k >= i) {
                        cplx tik = sT[k * QR_NB + i];
                        cplx t2 = q_cmul(tik, w[k]);
                        acc.x += t2.x; acc.y += t2.y;
                    }
                }
                cplx t3 = q_cmul(tau, acc);
                t = make_double2(-t3.x, -t3.y);
            } else if (i == C) {
                t = tau;
            }
            sT[C * QR_NB + i] = t;
            if (i == C) { sTau[C] = tau; sBeta[C] = beta; }
        }
        __syncthreads();
    }
};

// Factor the panel A[j0:n, j0:j0+NB]:  R (upper part) goes back into A, the reflectors into Vp (unit lower
// trapezoidal, zeros above the diagonal, same position as in A), T into Tp[NB*NB] (column major) and
// -T into Tn[NB*NB].
template<int RPT, int NT>
__global__ __launch_bounds__(NT) void k_qr_panel(cplx* __restrict__ A, int lda, int n, int j0,
                                                  cplx* __restrict__ Vp, cplx* __restrict__ Tp, cplx* __restrict__ Tn, size_t cs) {
    __shared__ double red[2][NT / 64][2 * QR_NB];
    CHAIN(A); CHAIN(Vp); CHAIN(Tp); CHAIN(Tn);
    __shared__ cplx sT[QR_NB * QR_NB];
    __shared__ cplx sTau[QR_NB];
    __shared__ double sBeta[QR_NB];
    __shared__ cplx sAlpha;
    QrPanelState<RPT, NT> s;
    s.red = red; s.sT = sT; s.sTau = sTau; s.sBeta = sBeta; s.sAlpha = &sAlpha;
    s.tid = threadIdx.x; s.lane = threadIdx.x & 63; s.wave = threadIdx.x >> 6;
    s.rows = n - j0;
    s.ncols = (n - j0 < QR_NB) ? (n - j0) : QR_NB;
    for (int i = threadIdx.x; i < QR_NB * QR_NB; i += NT) sT[i] = make_double2(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        int row = s.tid + r * NT;
#pragma unroll
        for (int c = 0; c < QR_NB; ++c) {
            // clamped address + select (a guarded load costs an exec-masked branch and an s_waitcnt vmcnt(0) each)
            const cplx t = A[(size_t)(j0 + min(c, s.ncols - 1)) * lda + (j0 + min(row, s.rows - 1))];
            s.a[r][c] = (row < s.rows && c < s.ncols) ? t : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    QrPanelStep<0>::run(s);
    // ---- write back ----
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        int row = s.tid + r * NT;
        if (row < s.rows) {
#pragma unroll
            for (int c = 0; c < QR_NB; ++c) {
                if (c >= s.ncols) continue;
                size_t off = (size_t)(j0 + c) * lda + (j0 + row);
                cplx x = s.a[r][c];
                if (row < c) {                 // strictly upper part of the top block: R
                    A[off] = x;
                    Vp[off] = make_double2(0.0, 0.0);
                } else if (row == c) {
                    A[off] = make_double2(sBeta[c], 0.0);
                    Vp[off] = make_double2(1.0, 0.0);
                } else {
                    A[off] = make_double2(0.0, 0.0);
                    Vp[off] = x;
                }
            }
        }
    }
    for (int i = threadIdx.x; i < QR_NB * QR_NB; i += NT) {
        cplx t = sT[i];
        Tp[i] = t;
        Tn[i] = make_double2(-t.x, -t.y);
    }
}


// Apply a block reflector to 16 columns of C per workgroup:  C <- (I - V op(T) V^H) C,  op(T) = T^H while
// factoring (Q_p^H on the trailing matrix), T while forming Q.  Tn holds -T.  One launch replaces the three
// small GEMMs (W = V^H C, W2 = -op(T) W, C += V W2), both big ones on the matrix cores
// (v_mfma_f64_16x16x4_f64, 4 real MFMAs per complex 16x16x4 step):
//   pass 1  W[i][j] = sum_row conj(V[row,i]) C[row,j]: V and C are streamed through LDS in 64-row chunks
//           (coalesced loads), every wave contracts 16 rows of the chunk, the four partial W are summed in LDS;
//   pass 2  C[row,j] += sum_k V[row,k] W2[k][j] as the transposed product (operand roles swapped) so that a lane
//           owns 4 columns of ONE row and the 16 lanes of a group own 16 consecutive rows: the V fragments and
//           the read-modify-write of C go straight to global memory, 256 contiguous bytes per group.
typedef double q_v4d __attribute__((ext_vector_type(4)));
template<bool TRANS_T>
// (256, 2): with 512 registers on offer the compiler parks the accumulators in AGPRs and copies them back and forth in the loops
__global__ __launch_bounds__(256, 2) void k_qr_apply(const cplx* __restrict__ Vp, int ldv, const cplx* __restrict__ Tn,
                                                   cplx* __restrict__ C, int ldc, int rows, int ncols, int nb, size_t cs) {
    __shared__ cplx sV[64][QR_NB + 1];
    __shared__ cplx sC[64][QR_NB + 1];
    __shared__ cplx sW[QR_NB][QR_NB + 1];      // W, then W2, as [k][j]
    __shared__ cplx sT[QR_NB][QR_NB + 1];      // -T as [i][k]
    __shared__ cplx sPart[4][QR_NB][QR_NB + 1];
    CHAIN(Vp); CHAIN(Tn); CHAIN(C);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int c0 = blockIdx.x * QR_NB;
    const int nc = min(QR_NB, ncols - c0);
    {
        int i = tid & 15, k = tid >> 4;
        sT[i][k] = (i < nb && k < nb) ? Tn[k * QR_NB + i] : make_double2(0.0, 0.0);
    }
    // ---- pass 1: W = V^H C ----
    q_v4d w_re = (q_v4d)(0.0), w_im = (q_v4d)(0.0);
    for (int r0 = 0; r0 < rows; r0 += 64) {
        __syncthreads();
        for (int idx = tid; idx < 64 * QR_NB; idx += 256) {
            int rr = idx & 63, cc = idx >> 6;
            int row = r0 + rr;
            // clamped addresses + select: guarded loads are issued one at a time (branch + s_waitcnt vmcnt(0) each)
            const int rowc = min(row, rows - 1);
            const cplx tv = Vp[(size_t)min(cc, nb - 1) * ldv + rowc];
            const cplx tc = C[(size_t)(c0 + min(cc, nc - 1)) * ldc + rowc];
            sV[rr][cc] = (row < rows && cc < nb) ? tv : make_double2(0.0, 0.0);
            sC[rr][cc] = (row < rows && cc < nc) ? tc : make_double2(0.0, 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int rl = wave * 16 + ks * 4 + l4;
            const cplx v = sV[rl][l15], x = sC[rl][l15];       // A(m = i, k) = conj(v), B(k, n = j) = x
            w_re = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, x.x, w_re, 0, 0, 0);
            w_re = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, x.y, w_re, 0, 0, 0);
            w_im = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, x.y, w_im, 0, 0, 0);
            w_im = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.y, x.x, w_im, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sPart[wave][l4 + 4 * r][l15] = make_double2(w_re[r], w_im[r]);   // D[m = l4 + 4r][n = l15]
    __syncthreads();
    const int wi = tid & 15, wj = tid >> 4;
    {
        cplx p0 = sPart[0][wi][wj], p1 = sPart[1][wi][wj], p2 = sPart[2][wi][wj], p3 = sPart[3][wi][wj];
        sW[wi][wj] = make_double2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    }
    __syncthreads();
    // ---- W2 = (-T)^(H) W ----
    {
        cplx a2 = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < QR_NB; ++k) {
            cplx t = TRANS_T ? make_double2(sT[k][wi].x, -sT[k][wi].y) : sT[wi][k];
            cplx w = sW[k][wj];
            a2.x += t.x * w.x - t.y * w.y;
            a2.y += t.x * w.y + t.y * w.x;
        }
        __syncthreads();
        sW[wi][wj] = a2;
    }
    __syncthreads();
    // ---- pass 2: C^T tile (16 columns x 16 rows) += W2^T V^T; lane: row = row0 + l15, columns c0 + l4 + 4r ----
    cplx w2[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2[ks] = sW[ks * 4 + l4][l15];           // A(m = j = l15, k)
    for (int row0 = wave * 16; row0 < rows; row0 += 64) {
        const int row = row0 + l15;
        const bool rok = row < rows;
        cplx vf[4], cv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 4 + l4;
            const cplx tv = Vp[(size_t)min(k, nb - 1) * ldv + min(row, rows - 1)];
            vf[ks] = (rok && k < nb) ? tv : make_double2(0.0, 0.0);                          // B(k, n = row)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = l4 + 4 * r;
            cv[r] = C[(size_t)(c0 + min(j, nc - 1)) * ldc + min(row, rows - 1)];             // only stored back where rok && j < nc
        }
        q_v4d d_re = (q_v4d)(0.0), d_im = (q_v4d)(0.0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            d_re = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[ks].x, vf[ks].x, d_re, 0, 0, 0);
            d_re = __builtin_amdgcn_mfma_f64_16x16x4f64(-w2[ks].y, vf[ks].y, d_re, 0, 0, 0);
            d_im = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[ks].x, vf[ks].y, d_im, 0, 0, 0);
            d_im = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[ks].y, vf[ks].x, d_im, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = l4 + 4 * r;                                       // D[m = j][n = row]
            if (rok && j < nc) C[(size_t)(c0 + j) * ldc + row] = make_double2(cv[r].x + d_re[r], cv[r].y + d_im[r]);
        }
    }
}

// The same update with the 16 columns of C held in REGISTERS between the two passes (rows <= 512): every wave owns a
// slab of rows, its lanes keep the slab in MFMA fragment order -- element e of lane (l15, l4) is C[slab + 4 e + l4,
// c0 + l15], which is at once the B fragment of pass 1 (k = row, n = column) and the accumulator layout of pass 2
// (m = row = l4 + 4 r, n = column) -- so C is read once and written once instead of read twice, and pass 1 needs no
// LDS staging or barrier per chunk.  V comes from global memory (all workgroups of a chain read the same panel: L2).
#define QR_MAXREF 4
// keeps loads on their side (compiler level: memory clobber; scheduler level: sched_barrier): the double buffers below are
// otherwise undone by hoisting every load of a pass to its top (32 quads of registers -> spills)
#define QR_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// makes the base pointer of the next loads depend on an accumulator of the MFMAs before it: the loads cannot be hoisted
// above those MFMAs, and the MFMAs after it cannot move above the loads
#define QR_TIE(acc, ptr) asm volatile("" : "+v"(acc), "+v"(ptr))
struct QrRefs { const cplx* V[QR_MAXREF]; const cplx* Tn[QR_MAXREF]; int nb[QR_MAXREF]; int n; };
// NCH = ceil(rows / 64) is a template parameter and every load has a clamped, always valid address: the unrolled code is
// straight-line, so the compiler issues the loads of a pass in batches instead of one guarded load + s_waitcnt vmcnt(0)
// per element (what the bounds-checked version compiled to: waves were parked half of their cycles).
template<bool TRANS_T, int NCH>
__global__ __launch_bounds__(256, 2) void k_qr_apply_reg(QrRefs refs, int ldv, cplx* __restrict__ C, int ldc, int rows, int ncols, size_t cs, int nb_chains) {
    __shared__ cplx sW[QR_NB][QR_NB + 1];      // W, then W2, as [k][j]
    __shared__ cplx sT[QR_NB][QR_NB + 1];      // -T as [i][k]
    __shared__ cplx sPart[4][QR_NB][QR_NB + 1];
    // all column blocks of a chain on ONE XCD (1-D grid, chains a multiple of 8): the reflector panels V, which every
    // workgroup of the chain reads twice per reflector, are then fetched into one L2 instead of eight (rocprofv3 FETCH_SIZE of
    // the trailing update before: 2.3x its algorithmic bytes, profiles/r02_pmc_traffic_b128_d32.json)
    int chain, cblk;
    xcd_chain_tile((ncols + QR_NB - 1) / QR_NB, nb_chains, chain, cblk);
    C = chain_ptr_i(C, cs, chain);
    constexpr int RW = 16 * NCH;                    // rows per wave
    constexpr int NE = 4 * NCH;                     // elements per lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int c0 = cblk * QR_NB;
    const int nc = min(QR_NB, ncols - c0);
    const int slab = wave * RW;
    const int rlast = rows - 1;
    const bool cok = l15 < nc;
    cplx* Ccol = C + (size_t)(c0 + (cok ? l15 : 0)) * ldc;
    cplx creg[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int row = slab + 4 * e + l4;
        const cplx t = nt_load(Ccol + min(row, rlast));       // C is streamed through once; the reflector panels stay in the L2
        creg[e] = (cok && row < rows) ? t : make_double2(0.0, 0.0);
    }
    // Up to four block reflectors are applied one after the other while C stays in registers (the later ones are the
    // next panels', whose rows above their own first row are zero in V): the trailing matrix is then read and written
    // once per GROUP of panels.
    const int wi = tid & 15, wj = tid >> 4;
    for (int rf = 0; rf < refs.n; ++rf) {
        // refs.V[rf] is a pointer picked from a by-value struct with a run-time index: the compiler cannot prove its address space and
        // every load through it became flat_load (48 in the NCH = 8 instantiation; flat loads count on vmcnt AND lgkmcnt, so they
        // also serialise with the LDS traffic of the reductions) -- say that it is global memory
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const cplx __attribute__((address_space(1)))* gcplx;
#else
        typedef const cplx* gcplx;             // host pass: the body is only parsed
#endif
        gcplx Vp = (gcplx)chain_ptr_i(refs.V[rf], cs, chain);
        gcplx Tn = (gcplx)chain_ptr_i(refs.Tn[rf], cs, chain);
        const int nb = refs.nb[rf];
        // per-iteration copies of the bounds the compiler cannot see through: otherwise the 32 clamped row indices and
        // 32 lane masks of the passes are hoisted out of this loop and kept alive across it (spills at NCH = 8)
        int rl = rlast, rw = rows;
        asm volatile("" : "+v"(rl), "+s"(rw));
        __syncthreads();                            // sT / sW / sPart of the previous reflector are no longer read
        {
            const cplx t = Tn[wj * QR_NB + wi];     // the panel kernel writes all 256 entries (zeros beyond its columns)
            sT[wi][wj] = (wi < nb && wj < nb) ? t : make_double2(0.0, 0.0);
        }
        const bool vok = l15 < nb;
        gcplx Vcol = Vp + (size_t)(vok ? l15 : 0) * ldv;
        // ---- pass 1: W = V^H C over this wave's slab; V in chunks of CH elements, the next chunk's loads in flight while
        //      the matrix cores work on the current one (explicit double buffer: left alone the compiler keeps ONE
        //      register quad for v and waits for every load right after issuing it) ----
        // 3M complex products (kernels_gemm.hip): A = conj(v) = (v.x, -v.y), B = x:  w_re = P1 = v.x x.x,  w_p2 = P2 = -v.y x.y,
        // w_im = P3 = (v.x - v.y)(x.x + x.y);  W = (P1 - P2, P3 - P1 - P2)
        q_v4d w_re = (q_v4d)(0.0), w_im = (q_v4d)(0.0), w_p2 = (q_v4d)(0.0);
        constexpr int CH = 4;                      // 8 with unmasked loads spills at NCH = 8 (creg alone is 128 registers)
        cplx va[CH], vb[CH];
        // (the row index of an element is re-derived from an opaque copy of slab + l4 at every use: left to itself the compiler keeps
        //  all 32 of them in registers across both passes -- with the 128 registers of creg that is what spilled at NCH = 8)
        auto loadv = [&](gcplx base, int e0, cplx (&dst)[CH]) {
            int rb = slab + l4;
            asm volatile("" : "+v"(rb));
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (e0 + i >= NE) continue;                                                   // NE need not be a multiple of CH
                const int row = rb + 4 * (e0 + i);
                // NO mask on the value: rows beyond the matrix meet creg = 0 (kept zero below), reflector columns >= nb give rows of
                // W that the zero-padded T annihilates -- the address is clamped, so what is loaded is finite.  A select here
                // (`ok ? load : 0`) compiles to a branch around the load and an s_waitcnt vmcnt(0) behind it: 68 of them in the
                // NCH = 8 instantiation, one per fragment -- the double buffer never had two loads in flight (round 3, ISA).
                dst[i] = base[min(row, rl)];                                                  // A(m = i, k) = conj(v)
            }
        };
        auto mac = [&](int e0, const cplx (&src)[CH]) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (e0 + i >= NE) continue;
                const cplx v = src[i];
                const cplx x = creg[e0 + i];                                                  // B(k, n = j)
                w_re = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, x.x, w_re, 0, 0, 0);
                w_p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.y, x.y, w_p2, 0, 0, 0);
                w_im = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x - v.y, x.x + x.y, w_im, 0, 0, 0);
            }
        };
        gcplx vbase = Vcol;
        loadv(vbase, 0, va);
        if (CH < NE) loadv(vbase, CH, vb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NE; c += 2 * CH) {
            mac(c, va);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 * CH < NE) { QR_TIE(w_re, vbase); loadv(vbase, c + 2 * CH, va); __builtin_amdgcn_sched_barrier(0); }
            if (c + CH < NE) { mac(c + CH, vb); __builtin_amdgcn_sched_barrier(0); }
            if (c + 3 * CH < NE) { QR_TIE(w_re, vbase); loadv(vbase, c + 3 * CH, vb); __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave][l4 + 4 * r][l15] = make_double2(w_re[r] - w_p2[r], (w_im[r] - w_re[r]) - w_p2[r]);   // D[m = l4 + 4r][n = l15]
        __syncthreads();
        {
            cplx p0 = sPart[0][wi][wj], p1 = sPart[1][wi][wj], p2 = sPart[2][wi][wj], p3 = sPart[3][wi][wj];
            sW[wi][wj] = make_double2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
        }
        __syncthreads();
        // ---- W2 = (-T)^(H) W ----
        {
            cplx a2 = make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < QR_NB; ++k) {
                cplx t = TRANS_T ? make_double2(sT[k][wi].x, -sT[k][wi].y) : sT[wi][k];
                cplx w = sW[k][wj];
                a2.x += t.x * w.x - t.y * w.y;
                a2.y += t.x * w.y + t.y * w.x;
            }
            __syncthreads();
            sW[wi][wj] = a2;
        }
        __syncthreads();
        // ---- pass 2: C tile (16 rows x 16 columns) += V W2 (3M: three zero-initialised accumulators, combined into C) ----
        cplx w2[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w2[ks] = sW[ks * 4 + l4][l15];           // B(k, n = column)
        cplx vfa[4], vfb[4];
        auto loadvf = [&](gcplx base, int t, cplx (&dst)[4]) {
            const int vrow = slab + 16 * t + l15;                                // A(m = row, k)
            const int vr = min(vrow, rl);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = ks * 4 + l4;
                dst[ks] = base[(size_t)min(k, nb - 1) * ldv + vr];       // unmasked: W2 rows >= nb are zero, rows beyond the matrix are re-zeroed in put()
            }
        };
        double w2s[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w2s[ks] = w2[ks].x + w2[ks].y;
        q_v4d d_re, d_im, d_p2;
        auto upd = [&](int t, const cplx (&vf)[4]) {
            d_re = (q_v4d)(0.0); d_im = (q_v4d)(0.0); d_p2 = (q_v4d)(0.0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                d_re = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[ks].x, w2[ks].x, d_re, 0, 0, 0);
                d_p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[ks].y, w2[ks].y, d_p2, 0, 0, 0);
                d_im = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[ks].x + vf[ks].y, w2s[ks], d_im, 0, 0, 0);
            }
        };
        auto put = [&](int t) {
            int rb = slab + l4;
            asm volatile("" : "+v"(rb));
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                    // D[m = l4 + 4r][n = l15]
                const bool rin = rb + 16 * t + 4 * r < rw;                           // rows beyond the matrix stay ZERO in creg (pass 1 of the next reflector relies on it)
                creg[4 * t + r].x = rin ? creg[4 * t + r].x + (d_re[r] - d_p2[r]) : 0.0;
                creg[4 * t + r].y = rin ? creg[4 * t + r].y + ((d_im[r] - d_re[r]) - d_p2[r]) : 0.0;
            }
        };
        gcplx pbase = Vp;
        loadvf(pbase, 0, vfa);
        if (NCH > 1) loadvf(pbase, 1, vfb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NCH; t += 2) {
            upd(t, vfa);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < NCH) { QR_TIE(d_re, pbase); loadvf(pbase, t + 2, vfa); __builtin_amdgcn_sched_barrier(0); }
            put(t);
            if (t + 1 < NCH) {
                upd(t + 1, vfb);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 3 < NCH) { QR_TIE(d_re, pbase); loadvf(pbase, t + 3, vfb); __builtin_amdgcn_sched_barrier(0); }
                put(t + 1);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int row = slab + 4 * e + l4;
        if (cok && row < rows) nt_store(Ccol + row, creg[e]);
    }
}

template<bool TRANS_T>
static void launch_apply_reg(const Launch& lc, dim3 grid, const QrRefs& r, int ldv, cplx* C, int ldc, int rows, int ncols) {
    if (lc.nb % 8 == 0) grid = dim3(grid.x * lc.nb, 1, 1);          // XCD-aware 1-D grid, see xcd_chain_tile
    switch ((rows + 63) / 64) {
        case 1: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 1>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 2: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 2>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 3: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 3>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 4: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 4>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 5: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 5>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 6: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 6>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        case 7: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 7>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
        default: hipLaunchKernelGGL((k_qr_apply_reg<TRANS_T, 8>), grid, dim3(256), 0, lc.st, r, ldv, C, ldc, rows, ncols, lc.cs, lc.nb); break;
    }
}

// ---------------------------------------------------------------------------------------------
// driver: A (n x n, ld n) -> R in place (strict lower part zeroed), Q explicit; V/T workspace
// ---------------------------------------------------------------------------------------------
static void gemm_small(const Launch& lc, int opA, int opB, const cplx* A, int lda, const cplx* B, int ldb, cplx* C, int ldc,
                       int M, int N, int K, int accumulate) {
    GemmArgs g = GemmArgs();
    g.A = A; g.lda = lda; g.opA = opA; g.B = B; g.ldb = ldb; g.opB = opB; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.Kmul = 1; g.accumulate = accumulate;
    launch_gemm(lc, g);
}

static void launch_panel(const Launch& lc, cplx* A, int n, int j0, cplx* V, cplx* Tp, cplx* Tn) {
    const int rows = n - j0;
    if (rows > 2560) {                                   // 1024 threads, 3 or 4 rows each (n <= 4096 enforced by the caller)
        if (rows <= 3072) hipLaunchKernelGGL((k_qr_panel<3, 1024>), dim3(1, 1, lc.nb), dim3(1024), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs);
        else              hipLaunchKernelGGL((k_qr_panel<4, 1024>), dim3(1, 1, lc.nb), dim3(1024), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs);
        return;
    }
    if (rows > 1024) {                                   // 512 threads, 3..5 rows each: still (mostly) register resident
        switch ((rows + 511) / 512) {
            case 3: hipLaunchKernelGGL((k_qr_panel<3, 512>), dim3(1, 1, lc.nb), dim3(512), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
            case 4: hipLaunchKernelGGL((k_qr_panel<4, 512>), dim3(1, 1, lc.nb), dim3(512), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
            default: hipLaunchKernelGGL((k_qr_panel<5, 512>), dim3(1, 1, lc.nb), dim3(512), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
        }
        return;
    }
    const int rpt = (rows + 255) / 256;
    switch (rpt) {
        case 1: hipLaunchKernelGGL((k_qr_panel<1, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
        case 2: hipLaunchKernelGGL((k_qr_panel<2, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
        case 3: hipLaunchKernelGGL((k_qr_panel<3, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
        case 4: hipLaunchKernelGGL((k_qr_panel<4, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, V, Tp, Tn, lc.cs); break;
        default: break;
    }
}

// returns number of kernel launches issued (for the profiling counters)
int run_qr(const Launch& lc, int n, cplx* A, cplx* Q, const QrWork& w) {
    int launches = 0;
    const int np = (n + QR_NB - 1) / QR_NB;
    const bool reg = n <= 512;                          // register-resident update kernel, panels applied in groups
    auto Tneg = [&](int p) { return w.T + (size_t)p * 2 * QR_NB * QR_NB + QR_NB * QR_NB; };
    auto Vat = [&](int pcol, int prow) { return w.V + (size_t)(pcol * QR_NB) * n + prow * QR_NB; };   // panel pcol's V from row block prow
    auto nbof = [&](int p) { return (n - p * QR_NB < QR_NB) ? (n - p * QR_NB) : QR_NB; };
    // the block reflectors of panels p0, p0 + step, ... (count of them, in that order) applied to ncols columns of M
    // starting at column col0, rows from row block rb (the first row of the lowest-numbered panel involved)
    auto apply = [&](bool transT, cplx* M, int rb, int col0, int ncols, int p0, int step, int count) {
        const int rows = n - rb * QR_NB;
        cplx* C = M + (size_t)col0 * n + rb * QR_NB;
        const dim3 grid((ncols + QR_NB - 1) / QR_NB, 1, lc.nb);
        if (w.apply_hooks) w.apply_hooks->begin(w.apply_hooks->user);
        if (reg) {
            QrRefs r;
            r.n = count;
            for (int i = 0; i < QR_MAXREF; ++i) { r.V[i] = nullptr; r.Tn[i] = nullptr; r.nb[i] = 0; }
            for (int i = 0; i < count; ++i) { const int p = p0 + i * step; r.V[i] = Vat(p, rb); r.Tn[i] = Tneg(p); r.nb[i] = nbof(p); }
            if (transT) launch_apply_reg<true>(lc, grid, r, n, C, n, rows, ncols);
            else        launch_apply_reg<false>(lc, grid, r, n, C, n, rows, ncols);
        } else {
            if (transT) hipLaunchKernelGGL((k_qr_apply<true>), grid, dim3(256), 0, lc.st, Vat(p0, rb), n, Tneg(p0), C, n, rows, ncols, nbof(p0), lc.cs);
            else        hipLaunchKernelGGL((k_qr_apply<false>), grid, dim3(256), 0, lc.st, Vat(p0, rb), n, Tneg(p0), C, n, rows, ncols, nbof(p0), lc.cs);
        }
        if (w.apply_hooks) w.apply_hooks->end(w.apply_hooks->user, 1);
        ++launches;
    };
    auto panel = [&](int p) {
        cplx* Tp = w.T + (size_t)p * 2 * QR_NB * QR_NB;
        launch_panel(lc, A, n, p * QR_NB, w.V, Tp, Tp + QR_NB * QR_NB);
        ++launches;
    };
    // ---- factorisation: Q_p^H = I - V T^H V^H applied to the trailing columns.  Groups of 4 (or 2) full panels: the
    //      columns of the group's later panels are brought up to date by small launches (look-ahead), the rest of the
    //      trailing matrix sees all reflectors of the group in ONE pass. ----
    for (int p = 0; p < np;) {
        const int j0 = p * QR_NB;
        const int left = n - j0;                                  // columns from this panel on
        if (reg && left > 4 * QR_NB) {                            // p .. p+3 are full and something remains after them
            panel(p);
            apply(true, A, p, j0 + QR_NB, QR_NB, p, 1, 1);                     // -> columns of p+1
            panel(p + 1);
            apply(true, A, p, j0 + 2 * QR_NB, 2 * QR_NB, p, 1, 2);             // -> columns of p+2, p+3
            panel(p + 2);
            apply(true, A, p + 2, j0 + 3 * QR_NB, QR_NB, p + 2, 1, 1);         // -> columns of p+3
            panel(p + 3);
            apply(true, A, p, j0 + 4 * QR_NB, left - 4 * QR_NB, p, 1, 4);      // -> everything behind the group
            p += 4;
        } else if (reg && left > 2 * QR_NB) {
            panel(p);
            apply(true, A, p, j0 + QR_NB, QR_NB, p, 1, 1);
            panel(p + 1);
            apply(true, A, p, j0 + 2 * QR_NB, left - 2 * QR_NB, p, 1, 2);
            p += 2;
        } else {
            panel(p);
            const int ntrail = left - nbof(p);
            if (ntrail > 0) apply(true, A, p, j0 + nbof(p), ntrail, p, 1, 1);
            p += 1;
        }
    }
    // ---- Q = H_0 H_1 ... applied to the identity, block reflectors in reverse order (zungqr): C <- (I - V T V^H) C ----
    if (!Q) return launches;                                      // the caller applies Q itself (run_qr_apply_q)
    launch_set_identity(lc, Q, n);
    ++launches;
    for (int p = np - 1; p >= 0;) {
        int cnt = 1;
        if (reg && nbof(p) == QR_NB) cnt = (p >= 3) ? 4 : (p >= 1 ? 2 : 1);
        const int lo = p - (cnt - 1);                             // H_p first, ..., H_lo last; rows / columns from block lo
        apply(false, Q, lo, lo * QR_NB, n - lo * QR_NB, p, -1, cnt);
        p -= cnt;
    }
    return launches;
}

// C <- Q C (trans == 0) or C <- Q^H C (trans != 0) for a dense n x n matrix C, Q from the reflectors the last run_qr left
// in w (zunmqr, left): n^3 multiply-adds, against 2/3 n^3 for forming Q plus n^3 for the product with it.
int run_qr_apply_q(const Launch& lc, int n, cplx* C, const QrWork& w, int trans) {
    int launches = 0;
    const int np = (n + QR_NB - 1) / QR_NB;
    const bool reg = n <= 512;
    auto nbof = [&](int p) { return (n - p * QR_NB < QR_NB) ? (n - p * QR_NB) : QR_NB; };
    // one launch: the reflectors of panels first, first + step, ... (cnt of them, in that order), rows from block lo
    auto group = [&](int first, int step, int cnt, int lo) {
        const int rows = n - lo * QR_NB;
        cplx* Cs = C + lo * QR_NB;                                 // rows from block lo, all columns
        const dim3 grid((n + QR_NB - 1) / QR_NB, 1, lc.nb);
        if (w.apply_hooks) w.apply_hooks->begin(w.apply_hooks->user);
        if (reg) {
            QrRefs r;
            r.n = cnt;
            for (int i = 0; i < QR_MAXREF; ++i) { r.V[i] = nullptr; r.Tn[i] = nullptr; r.nb[i] = 0; }
            for (int i = 0; i < cnt; ++i) {
                const int q = first + i * step;
                r.V[i] = w.V + (size_t)(q * QR_NB) * n + lo * QR_NB;
                r.Tn[i] = w.T + (size_t)q * 2 * QR_NB * QR_NB + QR_NB * QR_NB;
                r.nb[i] = nbof(q);
            }
            if (trans) launch_apply_reg<true>(lc, grid, r, n, Cs, n, rows, n);
            else       launch_apply_reg<false>(lc, grid, r, n, Cs, n, rows, n);
        } else {
            const cplx* Vq = w.V + (size_t)(first * QR_NB) * n + lo * QR_NB;
            const cplx* Tq = w.T + (size_t)first * 2 * QR_NB * QR_NB + QR_NB * QR_NB;
            if (trans) hipLaunchKernelGGL((k_qr_apply<true>), grid, dim3(256), 0, lc.st, Vq, n, Tq, Cs, n, rows, n, nbof(first), lc.cs);
            else       hipLaunchKernelGGL((k_qr_apply<false>), grid, dim3(256), 0, lc.st, Vq, n, Tq, Cs, n, rows, n, nbof(first), lc.cs);
        }
        if (w.apply_hooks) w.apply_hooks->end(w.apply_hooks->user, 1);
        ++launches;
    };
    if (trans) {                                                   // Q^H = H_{np-1}^H ... H_0^H: ascending, as in the factorisation
        for (int p = 0; p < np;) {
            int cnt = 1;
            if (reg) { cnt = (np - p >= 4) ? 4 : ((np - p >= 2) ? 2 : 1); if (nbof(p + cnt - 1) != QR_NB) cnt = 1; }
            group(p, 1, cnt, p);
            p += cnt;
        }
    } else {                                                       // Q = H_0 ... H_{np-1}: descending, as when forming Q
        for (int p = np - 1; p >= 0;) {
            int cnt = 1;
            if (reg && nbof(p) == QR_NB) cnt = (p >= 3) ? 4 : (p >= 1 ? 2 : 1);
            group(p, -1, cnt, p - (cnt - 1));
            p -= cnt;
        }
    }
    return launches;
}

// ---------------------------------------------------------------------------------------------
// right-hand triangular solve  Y R = C  (R upper triangular n x n), in place on C, blocked by NB:
//   diagonal blocks (32 wide):  C_J <- C_J R_JJ^-1  (one thread per row);  everything else: GEMMs (trsm_rec below)
// ---------------------------------------------------------------------------------------------
#define TRSM_NB 32
// trans: the triangular matrix is the conjugate transpose of the stored LOWER triangle; unit: its diagonal is 1 (not read)
__global__ __launch_bounds__(256) void k_trsm_block(cplx* __restrict__ C, int ldc, const cplx* __restrict__ R, int ldr,
                                                     int n, int j0, int nb, int trans, int unit, size_t cs) {
    __shared__ cplx sR[TRSM_NB][TRSM_NB + 1];
    CHAIN(C); CHAIN(R);
    for (int i = threadIdx.x; i < TRSM_NB * TRSM_NB; i += 256) {
        int r = i % TRSM_NB, c = i / TRSM_NB;
        cplx v = make_double2(0.0, 0.0);
        if (r < nb && c < nb && r <= c) {
            if (unit && r == c) v = make_double2(1.0, 0.0);
            else if (trans) { const cplx t = R[(size_t)(j0 + r) * ldr + (j0 + c)]; v = make_double2(t.x, -t.y); }
            else v = R[(size_t)(j0 + c) * ldr + (j0 + r)];
        }
        if (r == c && r < nb) {                 // the diagonal is stored as its reciprocal: one complex multiply per column and row instead of two divisions
            const double dn = v.x * v.x + v.y * v.y;
            v = make_double2(v.x / dn, -v.y / dn);
        }
        sR[r][c] = v;
    }
    __syncthreads();
    int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    cplx y[TRSM_NB];
    // all 32 loads go out together: clamped column instead of a select on the loaded value (which compiles to a branch around the
    // load and a wait behind it: 66 s_waitcnt vmcnt(0) for 34 loads before); columns >= nb are loaded but never used
#pragma unroll
    for (int c = 0; c < TRSM_NB; ++c) y[c] = C[(size_t)(j0 + min(c, nb - 1)) * ldc + row];
#pragma unroll
    for (int c = 0; c < TRSM_NB; ++c) {
        if (c < nb) {
            cplx acc = y[c];
#pragma unroll
            for (int k = 0; k < TRSM_NB; ++k) {
                if (k < c) {
                    cplx t = q_cmul(y[k], sR[k][c]);
                    acc.x -= t.x; acc.y -= t.y;
                }
            }
            y[c] = q_cmul(acc, sR[c][c]);
        }
    }
#pragma unroll
    for (int c = 0; c < TRSM_NB; ++c)
        if (c < nb) C[(size_t)(j0 + c) * ldc + row] = y[c];
}

__global__ void k_negate_copy_block(const cplx* __restrict__ R, int ldr, int rows, int cols, cplx* __restrict__ out, int ldo, size_t cs) {
    CHAIN(R); CHAIN(out);
    size_t total = (size_t)rows * cols;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % rows), j = (int)(idx / rows);
        cplx v = R[(size_t)j * ldr + i];
        out[(size_t)j * ldo + i] = make_double2(-v.x, -v.y);
    }
}

// recursive halving: the solve for columns [j0, j0 + len) once everything left of j0 has been subtracted.  The updates are then
// ONE product per level and half (N = K = 256, 128, 64, 32 at n = 512) instead of fifteen with N = 32 and K up to 480: the same
// multiply-adds on 64 x 64 tiles, which read half the operand bytes of the 32 x 32 ones.
static int trsm_rec(const Launch& lc, int n, const cplx* R, cplx* C, int j0, int len, int trans, int unit) {
    if (len <= TRSM_NB) {
        hipLaunchKernelGGL(k_trsm_block, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, C, n, R, n, n, j0, len, trans, unit, lc.cs);
        return 1;
    }
    const int h = ((len / 2 + TRSM_NB - 1) / TRSM_NB) * TRSM_NB;
    int launches = trsm_rec(lc, n, R, C, j0, h, trans, unit);
    // C[:, j0 + h .. j0 + len) -= Y[:, j0 .. j0 + h) R[j0 .. j0 + h, j0 + h .. j0 + len)
    GemmArgs g = GemmArgs();
    g.A = C + (size_t)j0 * n; g.lda = n; g.opA = 0; g.C = C + (size_t)(j0 + h) * n; g.ldc = n;
    if (trans) { g.B = R + (size_t)j0 * n + (j0 + h); g.ldb = n; g.opB = 1; }          // (L^H)[J1, J2] = conj(L[J2, J1])^T
    else       { g.B = R + (size_t)(j0 + h) * n + j0; g.ldb = n; g.opB = 0; }
    g.M = n; g.N = len - h; g.K = h; g.Kmul = 1; g.accumulate = 1; g.negate = 1; g.tag = 1;
    launch_gemm(lc, g);
    launches += 1;
    return launches + trsm_rec(lc, n, R, C, j0 + h, len - h, trans, unit);
}
int run_trsm_right_upper(const Launch& lc, int n, const cplx* R, cplx* C, const QrWork& w, int trans, int unit) {
    (void)w;
    return trsm_rec(lc, n, R, C, 0, n, trans, unit);
}

// ---------------------------------------------------------------------------------------------
// QR of LARGE matrices (n > 1024: O(3) lattices, n_g = 2304 at L = 24) WITHOUT tall panels.
//
// A 2304 x 16 complex panel is more than the register file of one CU holds: the register-resident panel kernel spills there
// (400 - 770 us per panel), a left-looking streaming panel is no faster (scripts/micro/attic/qr_panel_stream.hip.txt), and with 144
// panels + 288 block-reflector launches per factorisation a Householder QR of one 2304 x 2304 matrix takes ~ 90 ms of dependent
// small launches -- 60 % of a config-5 sweep.  The matrix cores want GEMMs instead:
//
//   block classical Gram-Schmidt with reorthogonalisation (BCGS2), block width 64, panels by Cholesky-QR, twice (CholQR2):
//       W1 = Q_prev^H A_j;  A_j -= Q_prev W1;  W2 = Q_prev^H A_j;  A_j -= Q_prev W2;        R[prev, j] = W1 + W2
//       G = A_j^H A_j = R1^H R1;  A_j <- A_j R1^-1;   G' = A_j^H A_j = R2^H R2;  Q_j = A_j R2^-1;    R[j, j] = R2 R1
//   -- 2 n^3 complex multiply-adds (Householder incl. forming Q: 8/3 n^3), all of them in k_zgemm launches that carry every chain.
//
// Why it is accurate HERE: the matrices factored are B-chains times scales, columns pre-pivoted by norm.  Gram-Schmidt and
// Cholesky are invariant under column scaling (rounding errors are relative per column), so what matters is the conditioning of
// the column-EQUILIBRATED matrix -- a product of s = 10 slice matrices and a unitary factor, kappa ~ 10^2 - 10^3, far from the
// 10^8 where Cholesky-QR's kappa^2 breaks down; the second pass of each stage restores orthogonality to rounding (BCGS2: Barlow &
// Smoktunowicz 2013; CholQR2: Yamamoto et al. 2015).  Checked like the Householder path: tests/test_gpu_parity.py (Q unitary to
// 1e-11, G against the reference's CPU construction at n_g = 2304 to 1e-10, dqmc_tuning::qr_variant = 2 forces it on the small fixtures).
// ---------------------------------------------------------------------------------------------
#define BGS_NB 64
#define CHOL_PIVOT_TOL 1e-12
// Cholesky factor of the nb x nb Hermitian block at (j0, j0) of Gm (upper triangle read), R upper with R^H R = G written back in
// place (strict lower part zeroed).  If R1m != nullptr its block at (j0, j0) is replaced by R R1 (second CholQR pass: R_jj = R2 R1).
// One workgroup of SIXTEEN waves per chain.  Thread (j, q) owns the rows i = q (mod 16) of column j and keeps them in REGISTERS for the
// whole factorisation; the only thing that travels through LDS is the pivot row: the wave that owns row K + 1 publishes it (unscaled, up
// to date) while it applies step K, into the buffer the other parity is not reading -- a step costs ONE barrier, one LDS write and six
// independent LDS reads, and at most four element updates per thread.
// Round 3 (four waves, the block in LDS, 84 us per launch = 12 % of a single O(3) L = 24 chain's sweep): the 64 steps were bound by the
// ~ 450 instructions a wave issued per step; the prologue waited for every global load of R1 on its own (load, s_waitcnt vmcnt(0),
// ds_write, sixteen times); the product R2 R1 of the second pass ran a dependent chain of LDS reads per element.  Now: all loads of
// the prologue in flight together, the product as a loop over t with broadcast reads of R2's column t (same summation order).
// Arithmetic per element is unchanged except 1 / sqrt(pivot): one rsqrt sequence instead of a square root followed by a division.
#define CHOL_NQ 16
__global__ __launch_bounds__(64 * CHOL_NQ) void k_chol64(cplx* __restrict__ Gm, int ld, int j0, int nb, cplx* __restrict__ R1m, int* __restrict__ err, size_t cs) {
    __shared__ cplx s[BGS_NB][BGS_NB + 1];
    __shared__ cplx s1[BGS_NB][BGS_NB + 1];
    __shared__ cplx prow[2][BGS_NB];
    CHAIN(Gm); CHAIN(R1m); CHAIN(err);
    const int j = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // q in a scalar register: the row tests below are scalar branches
    constexpr int NS = BGS_NB / CHOL_NQ;           // rows per thread
    cplx a[NS], r1[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int i = CHOL_NQ * t + q;
        const bool in = j < nb && i <= j;
        a[t] = in ? Gm[(size_t)(j0 + j) * ld + (j0 + i)] : make_double2(0.0, 0.0);
        r1[t] = (in && R1m) ? R1m[(size_t)(j0 + j) * ld + (j0 + i)] : make_double2(0.0, 0.0);
    }
    if (R1m) {
#pragma unroll
        for (int t = 0; t < NS; ++t) s1[CHOL_NQ * t + q][j] = r1[t];
    }
    if (q == 0) prow[0][j] = a[0];                  // row 0
    // Pivot test: the K-th pivot is the squared distance of column K from the span of the columns before it; relative to the
    // column's own squared norm it is >= 1 / kappa^2 of the column-equilibrated panel.  Below CHOL_PIVOT_TOL (kappa > 10^6) the
    // re-orthogonalising second pass can no longer be trusted to restore orthogonality to rounding (CholQR2 needs kappa^2 eps << 1):
    // the panel is flagged -- also when the pivot is not positive at all -- and the host redoes the factorisation with Householder
    // panels (udt_dev / green_qr, dqmc_context.hip).  Never papered over: the flag is per chain and counted (dqmc_get_schedule_info).
    double g_orig = 1.0;                            // the diagonal entry of column j as it came in (held by the thread with q == j mod 16)
#pragma unroll
    for (int t = 0; t < NS; ++t) if (CHOL_NQ * t + q == j && j < nb) g_orig = a[t].x;
    for (int K = 0; K < nb; ++K) {
        __syncthreads();                            // pivot row K is complete in prow[K & 1]; nobody still reads prow[(K + 1) & 1]
        const cplx* pr = prow[K & 1];
        // all LDS reads of the step first (the rows are wave-uniform addresses: broadcasts) -- interleaved with the publish below the
        // compiler must assume that the write aliases the reads and waits for every read on its own
        cplx pi[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) pi[t] = pr[CHOL_NQ * t + q];
        const cplx pj = pr[j];
        const double dkk = pr[K].x;
        if (j == K && (K & (CHOL_NQ - 1)) == q && err && !(dkk > CHOL_PIVOT_TOL * g_orig)) *err = 1;
        // 1 / sqrt in ONE sequence (the step's longest dependent chain); d = dkk / sqrt(dkk)
        const double dk = fmax(dkk, 1e-300), id = rsqrt(dk), d = dk * id;
        const cplx rkj = make_double2(pj.x * id, pj.y * id);
        cplx pub = make_double2(0.0, 0.0);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int i = CHOL_NQ * t + q;          // scalar
            if (i > K) {
                if (i <= j && j < nb) {
                    const cplx rki = make_double2(pi[t].x * id, pi[t].y * id);
                    a[t].x -= rki.x * rkj.x + rki.y * rkj.y;                   // conj(r_Ki) r_Kj
                    a[t].y -= rki.x * rkj.y - rki.y * rkj.x;
                }
                if (i == K + 1) pub = a[t];         // the next pivot row, up to date
            } else if (i == K) {
                if (j >= K && j < nb) a[t] = (j == K) ? make_double2(d, 0.0) : rkj;    // row K of R
            }
        }
        if (((K + 1) & (CHOL_NQ - 1)) == q) prow[(K + 1) & 1][j] = pub;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int i = CHOL_NQ * t + q;
        s[i][j] = a[t];
        if (j < nb && i < nb) Gm[(size_t)(j0 + j) * ld + (j0 + i)] = (i <= j) ? a[t] : make_double2(0.0, 0.0);
    }
    if (!R1m) return;
    __syncthreads();
    // (R2 R1)[i][j] = sum_{t = i .. j} R2[i][t] R1[t][j], t ascending: one pass over t, R1[t][j] read once per t, R2[i][t] broadcast
    cplx acc[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) acc[u] = make_double2(0.0, 0.0);
    for (int t = 0; t < nb; ++t) {
        const cplx b = s1[t][j];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = CHOL_NQ * u + q;          // scalar
            if (i <= t) {
                const cplx x = s[i][t];
                if (t <= j) {
                    acc[u].x += x.x * b.x - x.y * b.y;
                    acc[u].y += x.x * b.y + x.y * b.x;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int i = CHOL_NQ * u + q;
        if (j < nb && i <= j) R1m[(size_t)(j0 + j) * ld + (j0 + i)] = acc[u];
    }
}
// C[0:rows, 0:cols] += D[0:rows, 0:cols] (both with leading dimension ld)
__global__ void k_add_block(cplx* __restrict__ C, const cplx* __restrict__ D, int ld, int rows, int cols, size_t cs) {
    CHAIN(C); CHAIN(D);
    const size_t total = (size_t)rows * cols;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % rows), jj = (int)(idx / rows);
        const size_t o = (size_t)jj * ld + i;
        const cplx d = D[o];
        cplx c = C[o];
        c.x += d.x; c.y += d.y;
        C[o] = c;
    }
}
__global__ void k_zero(cplx* __restrict__ A, size_t count, size_t cs) {
    CHAIN(A);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < count; idx += (size_t)gridDim.x * blockDim.x) A[idx] = make_double2(0.0, 0.0);
}
// The Householder kernels rely on the parts of w.V / w.T they never write being zero (the strict upper part of every reflector
// panel, the padding of T); run_qr_bgs uses w.V as scratch.  Called before a Householder factorisation that follows a block
// Gram-Schmidt one on the same workspace (the fallback of udt_dev / green_qr).
void qr_reset_workspace(const Launch& lc, int n, const QrWork& w) {
    const int np = (n + QR_NB - 1) / QR_NB;
    hipLaunchKernelGGL(k_zero, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, w.V, (size_t)n * n, lc.cs);
    hipLaunchKernelGGL(k_zero, dim3(64, 1, lc.nb), dim3(256), 0, lc.st, w.T, (size_t)np * 2 * QR_NB * QR_NB, lc.cs);
}
// A (n x n, ld n) -> R in place (upper triangular, positive diagonal), Q explicit (n x n); w.V is scratch.  Returns the launch count.
int run_qr_bgs(const Launch& lc, int n, cplx* A, cplx* Q, const QrWork& w) {
    int launches = 0;
    const size_t n2 = (size_t)n * n;
    launch_copy(lc, A, Q, n2);                                     // the columns of the working copy become Q
    hipLaunchKernelGGL(k_zero, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, A, n2, lc.cs);
    launches += 2;
    cplx* S = w.V;
    auto gemm = [&](int opA, const cplx* Am, int opB, const cplx* Bm, cplx* Cm, int M, int N, int K, int sub) {
        GemmArgs g = GemmArgs();
        g.A = Am; g.lda = n; g.opA = opA; g.B = Bm; g.ldb = n; g.opB = opB; g.C = Cm; g.ldc = n;
        g.M = M; g.N = N; g.K = K; g.Kmul = 1; g.accumulate = sub; g.negate = sub; g.tag = 1;
        if (w.part && (size_t)M * N * 2 <= w.part_count) { g.part = w.part; g.part_count = w.part_count; }      // split-K scratch: the skinny Q^H A / Gram products and the long-K panel updates
        launch_gemm(lc, g);
        ++launches;
    };
    for (int j0 = 0; j0 < n; j0 += BGS_NB) {
        const int b = (n - j0 < BGS_NB) ? (n - j0) : BGS_NB;
        cplx* Qj = Q + (size_t)j0 * n;
        cplx* Rj = A + (size_t)j0 * n;                               // rows 0 .. j0 - 1 of R's block column
        if (j0 > 0) {
            gemm(1, Q, 0, Qj, Rj, j0, b, n, 0);                      // W1 = Q_prev^H A_j
            gemm(0, Q, 0, Rj, Qj, n, b, j0, 1);                      // A_j -= Q_prev W1
            gemm(1, Q, 0, Qj, S, j0, b, n, 0);                       // W2 = Q_prev^H A_j   (what the first pass left behind)
            gemm(0, Q, 0, S, Qj, n, b, j0, 1);                       // A_j -= Q_prev W2
            hipLaunchKernelGGL(k_add_block, dim3(64, 1, lc.nb), dim3(256), 0, lc.st, Rj, S, n, j0, b, lc.cs);
            ++launches;
        }
        gemm(1, Qj, 0, Qj, Rj + j0, b, b, n, 0);                     // G = A_j^H A_j into R's diagonal block
        hipLaunchKernelGGL(k_chol64, dim3(1, 1, lc.nb), dim3(64 * CHOL_NQ), 0, lc.st, A, n, j0, b, (cplx*)nullptr, w.err, lc.cs);
        launches += 1 + trsm_rec(lc, n, A, Q, j0, b, 0, 0);          // A_j <- A_j R1^-1
        gemm(1, Qj, 0, Qj, S + (size_t)j0 * n + j0, b, b, n, 0);     // second pass: G' = A_j^H A_j (close to the identity)
        hipLaunchKernelGGL(k_chol64, dim3(1, 1, lc.nb), dim3(64 * CHOL_NQ), 0, lc.st, S, n, j0, b, A, w.err, lc.cs);   // R2; R_jj = R2 R1
        launches += 1 + trsm_rec(lc, n, S, Q, j0, b, 0, 0);          // Q_j = A_j R2^-1
    }
    return launches;
}

// ---------------------------------------------------------------------------------------------
// glue for the UDT decomposition  Ms P = Q R  ->  (Q, d, T^H) resp. (T^H, d, Q)
// ---------------------------------------------------------------------------------------------
// W[:, perm[j]] = Ms[:, j] (or Ms^H when T != 0), Ms = diag(rowscale) M diag(colscale); perm == nullptr: identity
__global__ void k_udt_init(const cplx* __restrict__ M, int ldm, const double* colscale, const double* rowscale,
                           const int* __restrict__ perm, int transpose, cplx* __restrict__ W, int n, size_t cs) {
    CHAIN(M); CHAIN(colscale); CHAIN(rowscale); CHAIN(perm); CHAIN(W);
    size_t total = (size_t)n * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % n), j = (int)(idx / n);
        int mi = transpose ? j : i, mj = transpose ? i : j;
        cplx v = M[(size_t)mj * ldm + mi];
        double sc = 1.0;
        if (colscale) sc *= colscale[mj];
        if (rowscale) sc *= rowscale[mi];
        W[(size_t)(perm ? perm[j] : j) * n + i] = make_double2(v.x * sc, transpose ? -v.y * sc : v.y * sc);
    }
}

// d[k] = |R[k,k]|
__global__ void k_udt_diag(const cplx* __restrict__ R, int n, double* d, size_t cs) {
    CHAIN(R); CHAIN(d);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { cplx r = R[(size_t)k * n + k]; d[k] = sqrt(r.x * r.x + r.y * r.y); }
}

// what the triangular product A (D^-1 R P^T)^H needs instead of the explicit factor: 1/d and the inverse permutation
__global__ void k_udt_lazy(const double* __restrict__ d, const int* __restrict__ perm, int n, double* dinv, int* perm_inv, size_t cs) {
    CHAIN(d); CHAIN(perm); CHAIN(dinv); CHAIN(perm_inv);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { dinv[k] = 1.0 / d[k]; perm_inv[perm[k]] = k; }
}

// Tt[j, k] = conj(R[k, perm[j]]) / d[k]   (= (D^-1 R P^T)^H); zero where R is zero
__global__ void k_udt_tmat(const cplx* __restrict__ R, const double* __restrict__ d, const int* __restrict__ perm,
                           int n, cplx* __restrict__ Tt, size_t cs) {
    CHAIN(R); CHAIN(d); CHAIN(perm); CHAIN(Tt);
    size_t total = (size_t)n * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(idx % n), k = (int)(idx / n);           // Tt(j, k), column major: contiguous in j
        int c = perm[j];
        cplx v = make_double2(0.0, 0.0);
        if (c >= k) {
            cplx r = R[(size_t)c * n + k];
            double inv = 1.0 / d[k];
            v = make_double2(r.x * inv, -r.y * inv);
        }
        Tt[idx] = v;
    }
}

// Y[:, perm[j]] = X[:, j] * colscale[j]
__global__ void k_permute_scale_cols(const cplx* __restrict__ X, const double* colscale, const int* __restrict__ perm,
                                     int n, cplx* __restrict__ Y, size_t cs) {
    CHAIN(X); CHAIN(colscale); CHAIN(perm); CHAIN(Y);
    size_t total = (size_t)n * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % n), j = (int)(idx / n);
        cplx v = X[idx];
        double sc = colscale ? colscale[j] : 1.0;
        Y[(size_t)(perm ? perm[j] : j) * n + i] = make_double2(v.x * sc, v.y * sc);
    }
}

// scale splitting of the UdV singular scales: dmax_inv = 1/max(d,1), dmin = min(d,1)
__global__ void k_split_scales(const double* __restrict__ d, int n, double* dmax_inv, double* dmin, size_t cs) {
    CHAIN(d); CHAIN(dmax_inv); CHAIN(dmin);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        double x = d[k];
        dmax_inv[k] = (x > 1.0) ? 1.0 / x : 1.0;
        dmin[k] = (x > 1.0) ? 1.0 : x;
    }
}

// sv[k] = |R[k,k]| * max(d_r[k],1) * max(d_l[k],1): its log-sum is log|det G^-1| (the only thing the
// global moves use, detsdwopdim.cpp:3613-3620)
__global__ void k_logdet_vector(const cplx* __restrict__ R, const double* __restrict__ drmax_inv,
                                const double* __restrict__ dlmax_inv, int n, double* sv, size_t cs) {
    CHAIN(R); CHAIN(drmax_inv); CHAIN(dlmax_inv); CHAIN(sv);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        cplx r = R[(size_t)k * n + k];
        sv[k] = sqrt(r.x * r.x + r.y * r.y) / (drmax_inv[k] * dlmax_inv[k]);
    }
}

void launch_udt_init(const Launch& lc, const cplx* M, int ldm, const double* cs, const double* rs, const int* perm,
                     int transpose, cplx* W, int n) {
    hipLaunchKernelGGL(k_udt_init, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, M, ldm, cs, rs, perm, transpose, W, n, lc.cs);
}
void launch_udt_diag(const Launch& lc, const cplx* R, int n, double* d) {
    hipLaunchKernelGGL(k_udt_diag, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, R, n, d, lc.cs);
}
void launch_udt_lazy(const Launch& lc, const double* d, const int* perm, int n, double* dinv, int* perm_inv) {
    hipLaunchKernelGGL(k_udt_lazy, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, d, perm, n, dinv, perm_inv, lc.cs);
}
void launch_udt_tmat(const Launch& lc, const cplx* R, const double* d, const int* perm, int n, cplx* Tt) {
    hipLaunchKernelGGL(k_udt_tmat, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, R, d, perm, n, Tt, lc.cs);
}
void launch_permute_scale_cols(const Launch& lc, const cplx* X, const double* cs, const int* perm, int n, cplx* Y) {
    hipLaunchKernelGGL(k_permute_scale_cols, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, X, cs, perm, n, Y, lc.cs);
}
void launch_split_scales(const Launch& lc, const double* d, int n, double* dmax_inv, double* dmin) {
    hipLaunchKernelGGL(k_split_scales, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, d, n, dmax_inv, dmin, lc.cs);
}
void launch_logdet_vector(const Launch& lc, const cplx* R, const double* a, const double* b, int n, double* sv) {
    hipLaunchKernelGGL(k_logdet_vector, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, R, a, b, n, sv, lc.cs);
}
