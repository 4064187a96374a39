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
// Right-looking, panel width 32:
//   k_lu_panel          one workgroup per chain factors a (rows x 32) panel held in registers (one row per thread): per column an
//                       arg-max reduction, the interchange through LDS, the scaling and the rank-1 update of the rest of the panel.
//                       Thread 0 also keeps the row permutation and boils the 32 interchanges down to ONE gather list (which old
//                       row ends up where) so that nobody has to replay them one after the other.
//   k_lu_rowswap_trsm   32 lanes per column outside the panel: lane r loads top row r (512 contiguous bytes per column) and one
//                       displaced row, the gather is two lane shuffles, the unit lower 32 x 32 solve for the columns right of
//                       the panel a forward substitution across the 32 lanes.
//   k_zgemm             trailing update A22 -= L21 U12 (K = 32: half the read-modify-write traffic of 16-wide panels, which is
//                       what bounds this product).
#include "dqmc_internal.h"

#define LU_NB 32

// max with the lane(s) a DPP control selects; lanes the control leaves out contribute 0 (the keys are non-negative)
template<int CTRL, int ROWMASK>
__device__ __forceinline__ double lu_dpp_max(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return fmax(v, __hiloint2double(hi2, lo2));
}

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

template<int NT>
struct LuPanelState {
    static constexpr int NW = NT / 64;
    cplx a[LU_NB];               // one row of the panel; WHICH row (relative to the panel's first row) is `myrow`
    int myrow;                   // an interchange renames rows instead of moving registers: the thread that holds the pivot row
                                 // becomes row C, the thread that was row C takes the pivot's old place
    double* redv;                // [NW] per-wave maximum key
    cplx* sPiv;                  // [NB] the pivot row (columns of the panel)
    int* sPrev;                  // the pivot row's place before the interchange
    int* sP;                     // [NB] pivot row of every column (relative to the panel's first row)
    int tid, lane, wave, rows, ncols;

    template<int C>
    __device__ __forceinline__ void column() {
        if (C >= ncols) return;
        // ---- pivot search: largest |a|^2 in column C among the rows at or below the diagonal.  ONE max-reduction of a 64-bit key:
        //      the bit pattern of a non-negative double orders like its value, so the low 9 mantissa bits carry 511 - thread (a pivot
        //      within 2^-43 of the largest candidate is as good as the largest).  DPP steps inside the wave, LDS across waves. ----
        double key = 0.0;
        {
            const double m = a[C].x * a[C].x + a[C].y * a[C].y;
            const unsigned long long bits = ((unsigned long long)__double_as_longlong(m) & ~0x1FFull) | (unsigned long long)(511 - tid);
            if (myrow >= C && myrow < rows && m == m) key = __longlong_as_double((long long)bits);
        }
        key = lu_dpp_max<0x111, 0xf>(key);      // row_shr:1
        key = lu_dpp_max<0x112, 0xf>(key);      // row_shr:2
        key = lu_dpp_max<0x114, 0xf>(key);      // row_shr:4
        key = lu_dpp_max<0x118, 0xf>(key);      // row_shr:8 -> lane 15 of every row of 16
        key = lu_dpp_max<0x142, 0xa>(key);      // row_bcast:15
        key = lu_dpp_max<0x143, 0xc>(key);      // row_bcast:31 -> lane 63
        if (lane == 63) redv[wave] = key;
        __syncthreads();
        double best = redv[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) best = fmax(best, redv[w]);
        // the thread that holds the pivot row; a column without a usable entry (all NaN): the thread that is row C already
        const bool none = best == 0.0 && (__double_as_longlong(best) == 0);
        const int pt = 511 - (int)((unsigned long long)__double_as_longlong(best) & 0x1FFull);
        const bool mine = none ? (myrow == C) : (tid == pt);
        if (mine) {
#pragma unroll
            for (int c = C; c < LU_NB; ++c) sPiv[c] = a[c];
            *sPrev = myrow;
            sP[C] = myrow;
        }
        __syncthreads();
        {
            const int prev = *sPrev;
            if (mine) myrow = C;
            else if (myrow == C) myrow = prev;
        }
        // ---- multipliers and rank-1 update of the rest of the panel (pivot row: broadcast reads from LDS, in groups of 8 with a
        //      scheduling fence in between: left alone the compiler requests the whole row at once and needs 128 more registers) ----
#define LU_FENCE() __builtin_amdgcn_sched_barrier(0)
        const cplx piv = sPiv[C];
        const double dn = piv.x * piv.x + piv.y * piv.y;
        const cplx inv = dn > 0.0 ? make_double2(piv.x / dn, -piv.y / dn) : make_double2(0.0, 0.0);
        if (myrow > C && myrow < rows) {
            const cplx x = a[C];
            const cplx l = make_double2(x.x * inv.x - x.y * inv.y, x.x * inv.y + x.y * inv.x);
            a[C] = l;
#pragma unroll
            for (int c = C + 1; c < LU_NB; ++c) {
                const cplx u = sPiv[c];
                a[c].x -= l.x * u.x - l.y * u.y;
                a[c].y -= l.x * u.y + l.y * u.x;
                if ((c & 7) == 7) LU_FENCE();
            }
        }
    }
};

// gather list of a panel (per chain, LU_SWAP_INTS ints).  Slots 0 .. 31 are the top rows of the panel, slots 32 .. 32 + ndisp - 1
// the rows below them that an interchange touched:  [0] = ndisp (<= 32), [1 + c] = slot whose OLD content ends up in top row c,
// [33 + e] = row of displaced slot e (global), [65 + e] = slot whose old content ends up there.
template<int NT>
__global__ __launch_bounds__(NT) void k_lu_panel(cplx* __restrict__ A, int lda, int n, int j0, int* __restrict__ perm,
                                                  int* __restrict__ swaps, size_t cs) {
    __shared__ double redv[NT / 64];
    __shared__ cplx sPiv[LU_NB];
    __shared__ int sP[LU_NB], sPrev;
    __shared__ int rowof[2 * LU_NB], content[2 * LU_NB];          // thread 0's bookkeeping of the interchanges (epilogue)
    CHAIN(A); CHAIN(perm); CHAIN(swaps);
    LuPanelState<NT> s;
    s.redv = redv; s.sPiv = sPiv; s.sPrev = &sPrev; s.sP = sP;
    s.myrow = threadIdx.x;
    s.tid = threadIdx.x; s.lane = threadIdx.x & 63; s.wave = threadIdx.x >> 6;
    s.rows = n - j0;
    s.ncols = (n - j0 < LU_NB) ? (n - j0) : LU_NB;
    if (j0 == 0)
        for (int i = threadIdx.x; i < n; i += NT) perm[i] = i;
#pragma unroll
    for (int c = 0; c < LU_NB; ++c) {
        const cplx t = A[(size_t)(j0 + min(c, s.ncols - 1)) * lda + (j0 + min(s.tid, s.rows - 1))];
        s.a[c] = (s.tid < s.rows && c < s.ncols) ? t : make_double2(0.0, 0.0);
    }
    if (threadIdx.x < LU_NB) sP[threadIdx.x] = threadIdx.x;
    __syncthreads();
    LuPanelStep<0>::run(s);
    if (s.myrow < s.rows) {
#pragma unroll
        for (int c = 0; c < LU_NB; ++c)
            if (c < s.ncols) A[(size_t)(j0 + c) * lda + (j0 + s.myrow)] = s.a[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // net effect of the interchanges on the rows: content[slot] = the slot whose OLD content sits in slot's row afterwards
        int nslot = LU_NB;
        for (int c = 0; c < LU_NB; ++c) { rowof[c] = c; content[c] = c; }
        for (int c = 0; c < s.ncols; ++c) {
            const int p = sP[c];
            if (p == c) continue;
            int slot = -1;
            if (p < LU_NB) slot = p;
            else {
                for (int e = LU_NB; e < nslot; ++e) if (rowof[e] == p) slot = e;
                if (slot < 0) { slot = nslot++; rowof[slot] = p; content[slot] = slot; }
            }
            const int t = content[c]; content[c] = content[slot]; content[slot] = t;
            const int q0 = perm[j0 + c]; perm[j0 + c] = perm[j0 + p]; perm[j0 + p] = q0;
        }
        swaps[0] = nslot - LU_NB;
        for (int c = 0; c < LU_NB; ++c) swaps[1 + c] = content[c];
        for (int e = 0; e < LU_NB; ++e) {
            const bool used = LU_NB + e < nslot;
            swaps[1 + LU_NB + e] = j0 + (used ? rowof[LU_NB + e] : 0);
            swaps[1 + 2 * LU_NB + e] = used ? content[LU_NB + e] : 0;
        }
    }
}

// complex value of lane `src` of this lane's group of 32
__device__ __forceinline__ cplx lu_shfl32(cplx v, int src) {
    return make_double2(__shfl(v.x, src, 32), __shfl(v.y, src, 32));
}

// Every column outside the panel [j0, j0 + nbw): apply the panel's row interchanges; columns right of the panel also get
// U12 = L11^-1 A12 (unit lower block of the panel).  32 lanes per column, 8 columns per workgroup.
// Tneg (n x LU_NB, ld = lda): -U12^T, the second operand panel of the trailing update in the layout k_flush reads (both operand panels
// n x K with contiguous columns): Tneg[k * lda + c] = -U12[k, c], c counted from the first column right of the panel.
__global__ __launch_bounds__(256) void k_lu_rowswap_trsm(cplx* __restrict__ A, int lda, int n, int j0, int nbw,
                                                          const int* __restrict__ swaps, cplx* __restrict__ Tneg, size_t cs) {
    __shared__ cplx sL[LU_NB][LU_NB + 1];
    __shared__ int sS[LU_SWAP_INTS];
    CHAIN(A); CHAIN(swaps); CHAIN(Tneg);
    for (int i = threadIdx.x; i < LU_SWAP_INTS; i += 256) sS[i] = swaps[i];
    for (int i = threadIdx.x; i < LU_NB * LU_NB; i += 256) {
        const int r = i % LU_NB, c = i / LU_NB;
        sL[r][c] = (r > c && r < nbw && c < nbw) ? A[(size_t)(j0 + c) * lda + (j0 + r)] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int r = threadIdx.x & 31;
    const int ci = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = ci < n - nbw;                             // whole groups of 32 lanes are live or not
    const int col = live ? (ci < j0 ? ci : ci + nbw) : 0;
    cplx* Ac = A + (size_t)col * lda;
    const int ndisp = sS[0];
    // old contents: lane r holds top row r and displaced row r (clamped addresses; nothing is stored that is not live)
    const cplx vtop = Ac[j0 + min(r, nbw - 1)];
    const int drow = sS[1 + LU_NB + r];
    const cplx vext = Ac[r < ndisp ? drow : j0];
    // gather: what ends up in top row r and in displaced row r
    const int st = sS[1 + r], se = sS[1 + 2 * LU_NB + r];
    cplx ntop, next;
    {
        const cplx a0 = lu_shfl32(vtop, st & 31), a1 = lu_shfl32(vext, st & 31);
        ntop = st < LU_NB ? a0 : a1;
        const cplx b0 = lu_shfl32(vtop, se & 31), b1 = lu_shfl32(vext, se & 31);
        next = se < LU_NB ? b0 : b1;
    }
    if (col >= j0 + nbw) {
        // forward substitution with the unit lower block: after step k lanes > k have subtracted L[r][k] u_k
#pragma unroll
        for (int k = 0; k < LU_NB - 1; ++k) {
            const cplx uk = lu_shfl32(ntop, k);
            const cplx l = sL[r][k];                               // zero for r <= k
            ntop.x -= l.x * uk.x - l.y * uk.y;
            ntop.y -= l.x * uk.y + l.y * uk.x;
        }
    }
    if (live) {
        if (r < nbw && col >= j0 + nbw) Tneg[(size_t)r * lda + (col - (j0 + nbw))] = make_double2(-ntop.x, -ntop.y);
        if (r < nbw) Ac[j0 + r] = ntop;
        if (r < ndisp) Ac[drow] = next;
    }
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
int run_lu(const Launch& lc, int n, cplx* A, int* perm, int* swaps, cplx* tneg) {
    if (n > 512) return -1;
    int launches = 0;
    for (int j0 = 0; j0 < n; j0 += LU_NB) {
        const int nbw = (n - j0 < LU_NB) ? (n - j0) : LU_NB;
        const int rows = n - j0;
        if (rows <= 64)       hipLaunchKernelGGL((k_lu_panel<64>), dim3(1, 1, lc.nb), dim3(64), 0, lc.st, A, n, n, j0, perm, swaps, lc.cs);
        else if (rows <= 256) hipLaunchKernelGGL((k_lu_panel<256>), dim3(1, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, perm, swaps, lc.cs);
        else                  hipLaunchKernelGGL((k_lu_panel<512>), dim3(1, 1, lc.nb), dim3(512), 0, lc.st, A, n, n, j0, perm, swaps, lc.cs);
        ++launches;
        if (n - nbw > 0) {
            hipLaunchKernelGGL(k_lu_rowswap_trsm, dim3((n - nbw + 7) / 8, 1, lc.nb), dim3(256), 0, lc.st, A, n, n, j0, nbw, swaps, tneg, lc.cs);
            ++launches;
        }
        const int rest = n - j0 - nbw;
        // Trailing update A22 -= L21 U12, K = 32: a read-modify-write stream over A22 with a thin product riding on it -- the shape of
        // the delayed-update flush, so it runs on k_flush (operand fragments straight from global memory, tile read late, three
        // workgroups per CU) instead of the LDS-staged k_zgemm, whose pipeline never fills at K = 32 (round 3: these launches moved
        // 2.0 TB/s, profiles/r03_pmc_traffic_b128_d32.json "gemm_in_factorisation"; DQMC_LU_GEMM=1 keeps the old route)
        static const bool lu_gemm = dev_knob("DQMC_LU_GEMM") && atoi(dev_knob("DQMC_LU_GEMM")) != 0;
        if (rest > 0 && nbw == LU_NB && !lu_gemm) {
            launch_flush(lc, A + (size_t)j0 * n + (j0 + nbw), tneg, n, A + (size_t)(j0 + nbw) * n + (j0 + nbw), n, rest, nbw, nullptr, 1, /*tag=*/1);
            ++launches;
        } else if (rest > 0) {
            GemmArgs g = GemmArgs();
            g.A = A + (size_t)j0 * n + (j0 + nbw); g.lda = n; g.opA = 0;                 // L21
            g.B = A + (size_t)(j0 + nbw) * n + j0; g.ldb = n; g.opB = 0;                 // U12
            g.C = A + (size_t)(j0 + nbw) * n + (j0 + nbw); g.ldc = n;
            g.M = rest; g.N = rest; g.K = nbw; g.Kmul = 1; g.accumulate = 1; g.negate = 1; g.tag = 1;
            launch_gemm(lc, g);
            ++launches;
        }
    }
    return launches;
}
