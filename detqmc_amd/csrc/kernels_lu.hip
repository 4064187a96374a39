// LU factorisation with partial (row) pivoting on gfx950 (complex fp64, n <= 512) -- the inversion inside the QR-mode Green's
// function.  greenFromUdV (src/detmodel.h:769-818) needs the INVERSE of one n x n matrix per call; the reference gets it from a
// third SVD, round 1 of this build from a Householder QR (factor + triangular solve + application of Q^H: 2/3 + 1/2 + 1 n^3
// multiply-adds, the Q parts on the reflector kernel, which runs at 2/3 of the rate of the GEMM kernel).  The matrix that is
// inverted here has O(1) entries by construction (scale splitting, dqmc_context.hip green_qr), so plain partial pivoting is as
// accurate (numpy comparison against the SVD formula at L = 8, beta = 20: 1e-15 for both) and costs 1/3 + 1/2 + 1/2 n^3, all of
// it except the panels on k_zgemm.
//
//   P Z = L U,  P = the sequential row interchanges ipiv (LAPACK zgetrf convention), in place: unit lower L below the diagonal.
//
// Right-looking, panel width 16:
//   k_lu_panel          one workgroup per chain factors a (rows x 16) panel held in registers: per column an arg-max reduction,
//                       the interchange through LDS, the scaling and the rank-1 update of the rest of the panel.  Thread 0 also
//                       keeps the row permutation and boils the 16 interchanges down to ONE gather list (which old row ends
//                       up where) so that nobody has to replay them one after the other.
//   k_lu_rowswap_trsm   one thread per column outside the panel: gathers the <= 32 rows involved (all loads in flight together),
//                       solves with the unit lower 16 x 16 block for the columns right of the panel, stores.
//   k_zgemm             trailing update A22 -= L21 U12 (K = 16).
#include "dqmc_internal.h"

#define LU_NB 16

template<int C>
struct LuPanelStep {
    template<class S> __device__ static __forceinline__ void run(S& s) {
        s.template column<C>();
        LuPanelStep<C + 1>::run(s);
    }
};
template<>
struct LuPanelStep<LU_NB> {
    template<class S> __device__ static __forceinline__ void run(S&) {}
};

template<int RPT, int NT>
struct LuPanelState {
    static constexpr int NW = NT / 64;
    cplx a[RPT][LU_NB];          // rows tid + r NT (relative to the panel's first row)
    double* redv;                // [NW] per-wave maximum
    int* redi;                   // [NW] its row
    cplx* sPiv;                  // [NB] the pivot row (columns of the panel)
    cplx* sRow;                  // [NB] row C before the interchange
    int* sP;                     // [NB] pivot row of every column (relative to the panel's first row)
    int tid, lane, wave, rows, ncols;

    template<int C>
    __device__ __forceinline__ void column() {
        if (C >= ncols) return;
        // ---- pivot search: largest |a|^2 in column C at or below the diagonal, smallest row on ties ----
        double best = -1.0;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = tid + r * NT;
            const double m = a[r][C].x * a[r][C].x + a[r][C].y * a[r][C].y;
            if (row >= C && row < rows && (m > best)) { best = m; bidx = row; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bidx, off, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (lane == 0) { redv[wave] = best; redi[wave] = bidx; }
        __syncthreads();
        best = redv[0];
        int p = redi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const double ob = redv[w];
            const int oi = redi[w];
            if (ob > best || (ob == best && oi < p)) { best = ob; p = oi; }
        }
        if (p < C || p >= rows) p = C;                       // column of NaNs: leave the row where it is
        // ---- interchange rows C and p (all 16 columns of the panel) through LDS ----
        const int pt = p % NT, pr = p / NT;
        if (tid == pt) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) {
                cplx v = a[0][c];
#pragma unroll
                for (int r = 1; r < RPT; ++r) if (pr == r) v = a[r][c];
                sPiv[c] = v;
            }
        }
        if (tid == C) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c) sRow[c] = a[0][c];
            sP[C] = p;
        }
        __syncthreads();
        if (p != C) {
            if (tid == pt) {
#pragma unroll
                for (int c = 0; c < LU_NB; ++c) {
                    const cplx v = sRow[c];
#pragma unroll
                    for (int r = 0; r < RPT; ++r) if (pr == r) a[r][c] = v;
                }
            }
            if (tid == C) {
#pragma unroll
                for (int c = 0; c < LU_NB; ++c) a[0][c] = sPiv[c];
            }
        }
        // ---- multipliers and rank-1 update of the rest of the panel ----
        const cplx piv = sPiv[C];
        const double dn = piv.x * piv.x + piv.y * piv.y;
        const cplx inv = dn > 0.0 ? make_double2(piv.x / dn, -piv.y / dn) : make_double2(0.0, 0.0);
        cplx prow[LU_NB];
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) prow[c] = sPiv[c];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = tid + r * NT;
            if (row > C && row < rows) {
                const cplx x = a[r][C];
                const cplx l = make_double2(x.x * inv.x - x.y * inv.y, x.x * inv.y + x.y * inv.x);
                a[r][C] = l;
#pragma unroll
                for (int c = C + 1; c < LU_NB; ++c) {
                    a[r][c].x -= l.x * prow[c].x - l.y * prow[c].y;
                    a[r][c].y -= l.x * prow[c].y + l.y * prow[c].x;
                }
            }
        }
    }
};

// gather list of a panel (per chain, LU_SWAP_INTS ints): [0] = number of displaced rows below the top block (<= 16),
// [1 .. 16] = source row of top row c, [17 .. 32] = destination rows of the displaced ones, [33 .. 48] = their source rows
// (all row numbers global)
template<int RPT, int NT>
__global__ __launch_bounds__(NT) void k_lu_panel(cplx* __restrict__ A, int lda, int n, int j0, int* __restrict__ perm,
                                                  int* __restrict__ swaps, size_t cs) {
    __shared__ double redv[NT / 64];
    __shared__ int redi[NT / 64];
    __shared__ cplx sPiv[LU_NB], sRow[LU_NB];
    __shared__ int sP[LU_NB];
    CHAIN(A); CHAIN(perm); CHAIN(swaps);
    LuPanelState<RPT, NT> s;
    s.redv = redv; s.redi = redi; s.sPiv = sPiv; s.sRow = sRow; s.sP = sP;
    s.tid = threadIdx.x; s.lane = threadIdx.x & 63; s.wave = threadIdx.x >> 6;
    s.rows = n - j0;
    s.ncols = (n - j0 < LU_NB) ? (n - j0) : LU_NB;
    if (j0 == 0)
        for (int i = threadIdx.x; i < n; i += NT) perm[i] = i;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = s.tid + r * NT;
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) {
            const cplx t = A[(size_t)(j0 + min(c, s.ncols - 1)) * lda + (j0 + min(row, s.rows - 1))];
            s.a[r][c] = (row < s.rows && c < s.ncols) ? t : make_double2(0.0, 0.0);
        }
    }
    if (threadIdx.x < LU_NB) sP[threadIdx.x] = threadIdx.x;
    __syncthreads();
    LuPanelStep<0>::run(s);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = s.tid + r * NT;
        if (row < s.rows) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c)
                if (c < s.ncols) A[(size_t)(j0 + c) * lda + (j0 + row)] = s.a[r][c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // net effect of the interchanges on the rows outside this panel's columns: content[slot] = the row whose OLD content
        // sits in slot's row after all interchanges; slots 0..15 = the top rows, further slots = displaced rows
        int rowof[2 * LU_NB], content[2 * LU_NB];
        int nslot = LU_NB;
        for (int c = 0; c < LU_NB; ++c) { rowof[c] = c; content[c] = c; }
        for (int c = 0; c < s.ncols; ++c) {
            const int p = sP[c];
            if (p == c) continue;
            int slot = -1;
            if (p < LU_NB) slot = p;
            else {
                for (int e = LU_NB; e < nslot; ++e) if (rowof[e] == p) slot = e;
                if (slot < 0) { slot = nslot++; rowof[slot] = p; content[slot] = p; }
            }
            const int t = content[c]; content[c] = content[slot]; content[slot] = t;
            const int q0 = perm[j0 + c]; perm[j0 + c] = perm[j0 + p]; perm[j0 + p] = q0;
        }
        swaps[0] = nslot - LU_NB;
        for (int c = 0; c < LU_NB; ++c) swaps[1 + c] = j0 + content[c];
        for (int e = LU_NB; e < 2 * LU_NB; ++e) {
            swaps[1 + e] = j0 + (e < nslot ? rowof[e] : 0);
            swaps[1 + LU_NB + e] = j0 + (e < nslot ? content[e] : 0);
        }
    }
}

// Every column outside the panel [j0, j0 + nbw): apply the panel's row interchanges; columns right of the panel also get
// U12 = L11^-1 A12 (unit lower 16 x 16 block of the panel).  One thread per column.
__global__ __launch_bounds__(256) void k_lu_rowswap_trsm(cplx* __restrict__ A, int lda, int n, int j0, int nbw,
                                                          const int* __restrict__ swaps, size_t cs) {
    __shared__ cplx sL[LU_NB][LU_NB + 1];
    __shared__ int sS[LU_SWAP_INTS];
    CHAIN(A); CHAIN(swaps);
    for (int i = threadIdx.x; i < LU_SWAP_INTS; i += 256) sS[i] = swaps[i];
    for (int i = threadIdx.x; i < LU_NB * LU_NB; i += 256) {
        const int r = i % LU_NB, c = i / LU_NB;
        sL[r][c] = (r > c && r < nbw && c < nbw) ? A[(size_t)(j0 + c) * lda + (j0 + r)] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= n - nbw) return;
    const int col = ci < j0 ? ci : ci + nbw;
    cplx* Ac = A + (size_t)col * lda;
    const int ndisp = sS[0];
    cplx top[LU_NB], ext[LU_NB];
#pragma unroll
    for (int c = 0; c < LU_NB; ++c) top[c] = Ac[sS[1 + c]];
#pragma unroll
    for (int e = 0; e < LU_NB; ++e) ext[e] = Ac[sS[1 + 2 * LU_NB + e]];       // unused slots read row j0 (valid), never stored
    if (col >= j0 + nbw) {
#pragma unroll
        for (int c = 1; c < LU_NB; ++c) {
            cplx acc = top[c];
#pragma unroll
            for (int k = 0; k < LU_NB; ++k) {
                if (k < c) {
                    const cplx l = sL[c][k];
                    acc.x -= l.x * top[k].x - l.y * top[k].y;
                    acc.y -= l.x * top[k].y + l.y * top[k].x;
                }
            }
            top[c] = acc;
        }
    }
#pragma unroll
    for (int c = 0; c < LU_NB; ++c)
        if (c < nbw) Ac[j0 + c] = top[c];
#pragma unroll
    for (int e = 0; e < LU_NB; ++e)
        if (e < ndisp) Ac[sS[1 + LU_NB + e]] = ext[e];
}

// Y[:, j] = X[:, perm[j]] * colscale[perm[j]]
__global__ void k_gather_scale_cols(const cplx* __restrict__ X, const double* colscale, const int* __restrict__ perm,
                                    int n, cplx* __restrict__ Y, size_t cs) {
    CHAIN(X); CHAIN(colscale); CHAIN(perm); CHAIN(Y);
    const size_t total = (size_t)n * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % n), j = (int)(idx / n);
        const int src = perm[j];
        const cplx v = X[(size_t)src * n + i];
        const double sc = colscale ? colscale[src] : 1.0;
        Y[idx] = make_double2(v.x * sc, v.y * sc);
    }
}
void launch_gather_scale_cols(const Launch& lc, const cplx* X, const double* cs, const int* perm, int n, cplx* Y) {
    hipLaunchKernelGGL(k_gather_scale_cols, dim3(1024, 1, lc.nb), dim3(256), 0, lc.st, X, cs, perm, n, Y, lc.cs);
}

// A (n x n, ld n, n <= 512) -> L \ U in place; perm[i] = the row of the input that row i of L U is (P A = L U); swaps: workspace
// of LU_SWAP_INTS ints per chain.  Returns the number of launches, or -1 when n is outside what the panel kernel holds.
int run_lu(const Launch& lc, int n, cplx* A, int* perm, int* swaps) {
    if (n > 512) return -1;
    int launches = 0;
    for (int j0 = 0; j0 < n; j0 += LU_NB) {
        const int nbw = (n - j0 < LU_NB) ? (n - j0) : LU_NB;
        const int rows = n - j0;
        if (rows <= 256) hipLaunchKernelGGL((k_lu_panel<1, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, perm, swaps, lc.cs);
        else             hipLaunchKernelGGL((k_lu_panel<2, 256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, perm, swaps, lc.cs);
        ++launches;
        if (n - nbw > 0) {
            hipLaunchKernelGGL(k_lu_rowswap_trsm, dim3((n - nbw + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, nbw, swaps, lc.cs);
            ++launches;
        }
        const int rest = n - j0 - nbw;
        if (rest > 0) {
            GemmArgs g = GemmArgs();
            g.A = A + (size_t)j0 * n + (j0 + nbw); g.lda = n; g.opA = 0;                 // L21
            g.B = A + (size_t)(j0 + nbw) * n + j0; g.ldb = n; g.opB = 0;                 // U12
            g.C = A + (size_t)(j0 + nbw) * n + (j0 + nbw); g.ldc = n;
            g.M = rest; g.N = rest; g.K = nbw; g.Kmul = 1; g.accumulate = 1; g.negate = 1;
            launch_gemm(lc, g);
            ++launches;
        }
    }
    return launches;
}
