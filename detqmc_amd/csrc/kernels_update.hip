// Local Metropolis updates of one time slice, device resident (gfx950).
//
// Replaces DetSDW::updateInSlice / updateInSlice_delayed / updateInSliceThermalization
// (reference src/detsdwopdim.cpp:2428-2489, :3023-3175, :3294-3375) with proposeNewPhiBox
// (:3922-3931), deltaSPhi (:4186-4239) and get_delta_forsite (:3179-3289).
//
// Same Markov chain, same block structure (a block ends after delaySteps ACCEPTED updates,
// :3051-3056) and the same RNG consumption order (OPDIM draws per proposal, then one more only if
// prob <= 1, :3113), but the bookkeeping is re-derived for the GPU ("sub-matrix" form):
//
//   Accepted sites of the block so far: index set I (MSF rows per site), Delta_A = blockdiag(delta_l).
//   Effective Green's function  G' = G + G[:,I] W (G[I,:] - E_I),   W = Delta_A M_A^-1,
//   M_A = 1 + (1 - G[I,I]) Delta_A.            (Woodbury; identical to the reference's G + X Y)
//   For a candidate site c: S = G'[c,c] = G[c,c] + G[c,I] W G[I,c];   M' = 1 + (1 - S) delta';
//   ratio = det M' (== det of the reference's Mj, :3081-3084).  On acceptance W grows by block
//   bordering: p = W G[I,c], q = G[c,I] W, F = delta' M'^-1:
//       W <- [[ W + p F q,  p F ], [ F q, F ]]
//
//   => a decision touches only O(MSF*j) scattered entries of G and O((MSF j)^2) flops, never whole
//   rows/columns; the reference's per-proposal row/column corrections (1.6 M tiny zgemm per 4 sweeps)
//   disappear.  After the block, X = G[:,I] W and Gr = G[I,:] - E_I are formed by a parallel gather
//   kernel and G += X Gr is one MFMA GEMM (kernels_gemm.hip).
//
// The decision kernel is ONE workgroup of four wavefronts per chain: the chain of decisions is strictly sequential; per
// proposal there are two workgroup barriers (wave 0 does the scalar Metropolis arithmetic while waves 1-3 form p = W v and
// q = u W), cross-lane sums are DPP row reductions.  Nothing in the environment changes what this kernel computes: the
// only developer switch left is the phase timer, and it exists only in builds with -DDQMC_DECIDE_TIMING.
#include "dqmc_internal.h"
#include <mutex>

__device__ __forceinline__ cplx u_cfma(cplx a, cplx b, cplx c) {
    c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
    c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
    return c;
}
__device__ __forceinline__ cplx u_cmul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx u_csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// v[idx], idx in 0..3, as an OR of masked bit patterns (exact)
__device__ __forceinline__ double u_pick4d(int idx, double v0, double v1, double v2, double v3) {
    const long long m0 = idx == 0 ? -1ll : 0ll, m1 = idx == 1 ? -1ll : 0ll, m2 = idx == 2 ? -1ll : 0ll, m3 = idx == 3 ? -1ll : 0ll;
    return __longlong_as_double((__double_as_longlong(v0) & m0) | (__double_as_longlong(v1) & m1) |
                                (__double_as_longlong(v2) & m2) | (__double_as_longlong(v3) & m3));
}
__device__ __forceinline__ cplx u_pick4(int idx, cplx v0, cplx v1, cplx v2, cplx v3) {
    return make_double2(u_pick4d(idx, v0.x, v1.x, v2.x, v3.x), u_pick4d(idx, v0.y, v1.y, v2.y, v3.y));
}
__device__ __forceinline__ double u_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// phi + randRange(low, high) with every operation rounded on its own, as on the CPU reference
__device__ __forceinline__ double propose_component(double phi_old, double low, double high, double u) {
#pragma clang fp contract(off)
    double span = high - low;
    double t = span * u;
    double r = low + t;
    return phi_old + r;
}

// e^{sign dtau V} at one site (detsdwopdim.cpp:3188-3229); c0 / c1: the diagonal entries (0,0) = (2,2) and (1,1) = (3,3), both the
// cosh term while cdwU == 0; xs carries the factor coshTermCDWl otherwise
template<int MSF>
__device__ __forceinline__ void ev_matrix(cplx (&V)[MSF][MSF], double sign, const double* p, int opdim,
                                          double c0, double c1, double xs) {
#pragma unroll
    for (int a = 0; a < MSF; ++a)
#pragma unroll
        for (int b = 0; b < MSF; ++b) V[a][b] = make_double2(0.0, 0.0);
    double p0 = p[0], p1 = opdim > 1 ? p[1] : 0.0;
    cplx bx = make_double2(sign * p0 * xs, -sign * p1 * xs);
    cplx bcx = make_double2(sign * p0 * xs, sign * p1 * xs);
    V[0][0] = make_double2(c0, 0.0);
    V[1][1] = make_double2(c1, 0.0);
    V[0][1] = bx;
    V[1][0] = bcx;
    if (MSF == 4) {
        double ax = sign * p[2] * xs;
        V[2][2] = make_double2(c0, 0.0);
        V[3][3] = make_double2(c1, 0.0);
        V[0][3] = make_double2(ax, 0.0);
        V[3][0] = make_double2(ax, 0.0);
        V[1][2] = make_double2(-ax, 0.0);
        V[2][1] = make_double2(-ax, 0.0);
        V[3][2] = bx;
        V[2][3] = bcx;
    }
}

// determinant and inverse of a small complex matrix by Gauss-Jordan with partial pivoting
template<int MSF>
__device__ __forceinline__ cplx small_det_inv(const cplx (&Min)[MSF][MSF], cplx (&Inv)[MSF][MSF]) {
    if (MSF == 2) {
        cplx det = u_csub(u_cmul(Min[0][0], Min[1][1]), u_cmul(Min[0][1], Min[1][0]));
        double dn = det.x * det.x + det.y * det.y;
        cplx idet = make_double2(det.x / dn, -det.y / dn);
        Inv[0][0] = u_cmul(Min[1][1], idet);
        Inv[1][1] = u_cmul(Min[0][0], idet);
        Inv[0][1] = u_cmul(make_double2(-Min[0][1].x, -Min[0][1].y), idet);
        Inv[1][0] = u_cmul(make_double2(-Min[1][0].x, -Min[1][0].y), idet);
        return det;
    }
    cplx A[MSF][MSF];
#pragma unroll
    for (int i = 0; i < MSF; ++i)
#pragma unroll
        for (int j = 0; j < MSF; ++j) {
            A[i][j] = Min[i][j];
            Inv[i][j] = make_double2(i == j ? 1.0 : 0.0, 0.0);
        }
    cplx det = make_double2(1.0, 0.0);
#pragma unroll
    for (int c = 0; c < MSF; ++c) {
        int piv = c;
        double best = A[c][c].x * A[c][c].x + A[c][c].y * A[c][c].y;
#pragma unroll
        for (int r = c + 1; r < MSF; ++r) {
            double v = A[r][c].x * A[r][c].x + A[r][c].y * A[r][c].y;
            if (v > best) { best = v; piv = r; }
        }
        // row swap with compile-time indices only (a run-time row index would push the matrices into scratch memory)
#pragma unroll
        for (int r = c + 1; r < MSF; ++r) {
            const bool sw = (piv == r);
#pragma unroll
            for (int j = 0; j < MSF; ++j) {
                cplx t = A[c][j], u = A[r][j];
                A[c][j] = sw ? u : t; A[r][j] = sw ? t : u;
                t = Inv[c][j]; u = Inv[r][j];
                Inv[c][j] = sw ? u : t; Inv[r][j] = sw ? t : u;
            }
        }
        if (piv != c) det = make_double2(-det.x, -det.y);
        cplx pv = A[c][c];
        det = u_cmul(det, pv);
        double dn = pv.x * pv.x + pv.y * pv.y;
        cplx ipv = make_double2(pv.x / dn, -pv.y / dn);
#pragma unroll
        for (int j = 0; j < MSF; ++j) { A[c][j] = u_cmul(A[c][j], ipv); Inv[c][j] = u_cmul(Inv[c][j], ipv); }
#pragma unroll
        for (int r = 0; r < MSF; ++r) {
            if (r == c) continue;
            cplx f = A[r][c];
#pragma unroll
            for (int j = 0; j < MSF; ++j) {
                A[r][j] = u_csub(A[r][j], u_cmul(f, A[c][j]));
                Inv[r][j] = u_csub(Inv[r][j], u_cmul(f, Inv[c][j]));
            }
        }
    }
    return det;
}

// wavefront sum by DPP row operations (total broadcast through lane 63)
template<int CTRL, int ROWMASK>
__device__ __forceinline__ double u_dpp_add(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double u_wave_total(double v) {
    v = u_dpp_add<0xB1, 0xf>(v);
    v = u_dpp_add<0x4E, 0xf>(v);
    v = u_dpp_add<0x114, 0xf>(v);
    v = u_dpp_add<0x118, 0xf>(v);
    v = u_dpp_add<0x142, 0xa>(v);
    v = u_dpp_add<0x143, 0xc>(v);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// ---- rotate / scale proposals of the O(3) model (detsdwopdim.cpp:3934-4170), arithmetic in the reference's order, no fma contraction.
// The transcendental functions (sincos, pow, log) are the device library's: they agree with glibc's to an ulp or two, so a proposed
// field agrees with the reference's to ~1e-16 relative rather than bit for bit (tests: same decisions, fields to 1e-13).
// unit vector of the new direction: cone around the old direction v, cos(theta) = ct, azimuth ph (proposeRandomRotatedVector<3>, :3945-3992)
__device__ __forceinline__ void rot3_unit(const double (&v)[3], double ct, double ph, double& r_out, double (&out)[3]) {
#pragma clang fp contract(off)
    const double x = v[0], y = v[1], z = v[2];
    const double x2 = x * x, y2 = y * y, z2 = z * z;
    const double r2 = x2 + y2 + z2;
    const double r = sqrt(r2);
    const double st = sqrt(1.0 - ct * ct);
    double sp, cp;
    sincos(ph, &sp, &cp);
    const double x2n = x2 / r2, y2n = y2 / r2;
    const double xn = x / r, yn = y / r, zn = z / r;
    out[0] = (st / (x2n + y2n)) * ((x2n * zn + y2n) * cp + (zn - 1) * xn * yn * sp) + xn * ct;
    out[1] = (st / (x2n + y2n)) * ((zn - 1) * xn * yn * cp + (x2n + y2n * zn) * sp) + yn * ct;
    out[2] = -st * (xn * cp + yn * sp) + zn * ct;
    r_out = r;
}

// Everything the decision for ONE candidate site needs from global memory is loaded one candidate ahead
// (while the previous decision is being computed), so the L2 round trip of the scattered G entries and of
// the field values never sits on the critical path of the sequential chain.  The ~44 scalars (uniforms, field
// values, cosh/sinh, G[c,c], G[c,prev], G[prev,c]) are gathered by ONE wavefront with ONE load instruction --
// lane l loads item l -- and handed to the other waves through LDS: the same scalars loaded by every lane of
// every wave (wave-uniform vector loads) cost ~40 x 4 trips through the address coalescer per proposal.
// CDW 0: cdwU == 0.  CDW 1: phi proposals with the cdw terms of the site in e^{+-dtau V}.  CDW 2: the second pass over the slice
// (detsdwopdim.cpp:2474-2485) -- proposeNewCDWl (:4173-4182): ONE uniform picks the new l, phi stays, probSPhi = 1 and the ratio
// cdwl_gamma(new) / cdwl_gamma(old) joins the acceptance probability (:3110); its acceptance ratio is discarded.
// PROP (round 4; O(3) only, spinProposalMethod != box, detsdwopdim.cpp:3934-4170): 0 box, 1 proposeRotatedPhi (two uniforms: cos(theta) in
// [angleDelta, 1] and the azimuth), 2 proposeScaledPhi (|phi|^3 Gaussian around its old value: polar Box-Muller on the replica's stream,
// normaldistribution.h -- a VARIABLE number of uniforms, the second value of a pair kept for the next proposal; not positive => the
// proposal is dropped without an acceptance draw), 3 proposeRotatedScaledPhi (the Gaussian draw, then the two uniforms of the rotation).
// The uniforms of such a proposal come from a window of UNIW values that starts at the cursor the candidate's prefetch was issued
// with (exact: the decision before it is still open, but a window is long enough for both outcomes); beyond it: direct loads.
// adapt_what (thermalisation, end of the slice): 0 ADAPT_BOX, 1 ADAPT_ROTATE, 2 ADAPT_SCALE (+ 4: adaptScaleVariance), :3299-3375.
// NT threads per workgroup: 256, or 512 (O(1) / O(2); O(3) needs 256 registers per thread) with the waves specialised -- a wave of its own
// for the look-ahead loads, six for the p / q products and the rows of W.  512 is the faster kernel for a chain (single chain 6.8 -> 7.25
// sweeps/s, the decision launches of 128 chains alone on the GPU -13 %), but its 8 waves and 380 registers per SIMD push the other contexts'
// flush / GEMM waves off the CU: four contexts of 128 chains run 1.7 % SLOWER with it (same-box A/B, profiles/r04_decide_512.json).  The
// launcher therefore takes 512 for small batches only.
template<int OPDIM, int CDW, int PROP, int NT>
__global__ __launch_bounds__(NT) void k_update_decide(DevModel dm, DevUpdateState* us, const double* __restrict__ uni,
                                                       const cplx* __restrict__ Gfull, cplx* __restrict__ Wout,
                                                       int k, int first, int thermal, size_t cs,
                                                       const cplx* __restrict__ Gwin, int winP, int adapt_what, int reset_nd) {
    static_assert(PROP == 0 || (OPDIM == 3 && CDW != 2), "rotate / scale proposals belong to the phi pass of the O(3) model");
    constexpr int MSF = (OPDIM == 3) ? 4 : 2;
    static_assert(NT == 256 || (NT == 512 && OPDIM != 3), "launch shape");
    constexpr int NW = NT / 64;
    // Roles of the waves.  Wave 0: the scalar Metropolis arithmetic of a proposal (D), the border strips of W.  FETCH_WAVE: the loads of the
    // next candidate and their hand-over through LDS (A, C) -- with eight waves a wave of its own, which does this while the others
    // update W, with four waves wave 0.  Waves E_FIRST .. NW - 1: p = W v, q = u W (E) and the rows of W11.
    constexpr int FETCH_WAVE = (NW >= 8) ? 1 : 0, E_FIRST = FETCH_WAVE + 1, NWE = NW - E_FIRST;
    constexpr int U_FIRST = (NW >= 8) ? E_FIRST : 0, NWU = NW - U_FIRST;      // the waves that walk the rows of W11 (four waves: all of them)
    constexpr int SLOTS = MSF * DQMC_MAX_WDIM / 64;
    constexpr int NPROP = (CDW == 2) ? 1 : OPDIM;      // uniforms a BOX proposal draws
    constexpr int UNIW = 16;                           // PROP != 0: prefetched window of uniforms
    dm = chain_model(dm, cs); CHAIN(us); CHAIN(uni); CHAIN(Gfull); CHAIN(Wout); CHAIN(Gwin);
    dm.r = us->r;                         // the exchange parameter differs between the chains of a batch
    const int N = dm.N, D = dm.D, m = dm.m;
    const int WD = MSF * D;
    extern __shared__ cplx smem[];
    const int WS = WD + 1;                // row stride of W in LDS: odd, so rows do not start on the same banks
    cplx* W = smem;                       // [WD][WS] row-major: W[i*WS + i']
    cplx* su2 = W + WD * WS;              // u[a][i]  = G[c_a, I_i]      2 x [MSF][WD]  (ping-pong per proposal)
    cplx* sv2 = su2 + 2 * MSF * WD;       // v[i][b]  = G[I_i, c_b]      2 x [WD][MSF]
    cplx* sp = sv2 + 2 * WD * MSF;        // p = W v                     [WD][MSF]
    cplx* sq = sp + WD * MSF;             // q = u W                     [MSF][WD]
    double* sphi = (double*)(sq + MSF * WD);   // phi of slice k, [OPDIM][N]: all field reads and writes of the loop
    __shared__ int isite[DQMC_MAX_WDIM];       //   go here, the accepted values reach global memory after the loop
    __shared__ double sacc[DQMC_MAX_WDIM][OPDIM + 2 > 3 ? OPDIM + 2 : 3];   // accepted: new phi, cosh, sinh (CDW 2: l, its cosh, sinh)
    const int tid = threadIdx.x;          // the decision (S, det, accept) is formed redundantly by every wave,
    const int lane = tid & 63;            // the vector parts (p, q, W update) are split over the waves
    const int wave = tid >> 6;
    const int L = dm.L;

    int site = first ? 0 : us->site_cursor;
    // Every G entry a launch reads has both indices among the sites of its proposal window [site, site + pbudget) (x the MSF bands).
    // winP > 0: those entries come from the compact copy k_update_window made of that window -- already holding the previous block's
    // update, whose flush over the whole of G may still be running on the second stream (pipelined update, dqmc_update_slice).
    // Index of (site s, band a): (s - goff) + a * gNB, leading dimension gld.
    const cplx* G = winP > 0 ? Gwin : Gfull;
    const int goff = winP > 0 ? site : 0, gNB = winP > 0 ? winP : dm.N, gld = winP > 0 ? MSF * winP : dm.ng;
    int acc_count = first ? 0 : us->acc_count;
    int done = first ? 0 : us->slice_done;
    if (done || site >= N) {
        if (tid == 0) { us->block_j = 0; if (first) { us->site_cursor = site; us->acc_count = 0; us->slice_done = done; } }
        return;
    }
    unsigned long long cur = us->pub.rng_consumed;
    const unsigned long long avail = us->pub.rng_avail;
    const double phiDelta = us->pub.phiDelta;
    int err = us->pub.error;
    double* phik = dm.phi + (size_t)k * OPDIM * N;
    const int kEarlier = (k > 1) ? k - 1 : m;     // PeriodicChainNearestNeighbors<1> over slices 1..m
    const int kLater = (k < m) ? k + 1 : 1;
    const double* phiE = dm.phi + (size_t)kEarlier * OPDIM * N;
    const double* phiL = dm.phi + (size_t)kLater * OPDIM * N;
    const double* coshK = dm.coshT + (size_t)k * N;
    const double* sinhK = dm.sinhT + (size_t)k * N;

    // ---- scalar items of one candidate, as doubles (lane l of wave 0 loads item l, l + 64, ...) ----
    constexpr int O_UNI = 0;                       // OPDIM + 2 uniforms: the decision before may or may not have
    constexpr int O_TL = O_UNI + (PROP == 0 ? OPDIM + 2 : UNIW);   //   consumed one more than the guess the loads were issued with
    constexpr int O_TE = O_TL + OPDIM;             // phi(later slice), phi(earlier slice)
    constexpr int O_CH = O_TE + OPDIM;
    constexpr int O_SH = O_CH + 1;
    constexpr int O_CC = O_SH + 1;                 // CDW != 0: coshTermCDWl, sinhTermCDWl, l of the candidate site
    constexpr int O_SC = O_CC + 1;
    constexpr int O_CL = O_SC + 1;
    constexpr int GBLK = 2 * MSF * MSF;            // one MSF x MSF complex block; blocks are aligned to their size so
    constexpr int O_GCC = (O_CL + GBLK) / GBLK * GBLK;   // none straddles a group of 64 lanes:  G[c rows, c cols]
    constexpr int O_GNP = O_GCC + 2 * MSF * MSF;   // G[c_this rows, c_prev cols]
    constexpr int O_GPN = O_GNP + 2 * MSF * MSF;   // G[c_prev rows, c_this cols]
    constexpr int NITEMS = O_GPN + 2 * MSF * MSF;
    constexpr int NIT = (NITEMS + 63) / 64;
    static_assert(O_GCC % 2 == 0, "complex items must be 16-byte aligned in LDS");
    __shared__ __attribute__((aligned(16))) double scand[NIT * 64];
    __shared__ __attribute__((aligned(16))) double sdec[4 * MSF * MSF + 4];   // wave 0 -> all: delta, G[c,c], exp(-dS), uniform, (CDW 2: null-proposal flag)
    __shared__ cplx salg[MSF == 4 ? NW * 80 : 1];                              // O(3): wave-private scratch of the 4 x 4 algebra (S, M', cofactors, M'^-1, F)

    auto neighbours = [&](int s, int (&nbr)[4]) {
        // neighbortable.h:34-36 (XPLUS, XMINUS, YPLUS, YMINUS), computed: a table look-up would put a dependent
        // memory round trip in front of the field loads
        const int sx = s % L, sy = s / L;
        nbr[0] = sy * L + (sx + 1 == L ? 0 : sx + 1);
        nbr[1] = sy * L + (sx == 0 ? L - 1 : sx - 1);
        nbr[2] = (sy + 1 == L ? 0 : sy + 1) * L + sx;
        nbr[3] = (sy == 0 ? L - 1 : sy - 1) * L + sx;
    };
    // issue all loads for candidate `s` (wave 0 only); `prev` = the site whose decision is still open (-1: none),
    // `curGuess` = RNG cursor if that decision consumes no acceptance uniform, nIknown = MSF * (#accepted so far)
    auto fetch = [&](double (&pre)[NIT], cplx (&pu)[SLOTS], cplx (&pv)[SLOTS], int s, int prev,
                     unsigned long long curGuess, int nIknown) {
        if (wave != FETCH_WAVE) return;
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int item = lane + 64 * q;
            const double* addr = nullptr;
            double dflt = 0.0;
            if (item < O_TL) {
                unsigned long long idx = curGuess + item;
                if (idx < avail) addr = uni + idx;
                dflt = 0.5;
            } else if (item < O_TE) {
                addr = phiL + (item - O_TL) * N + s;
            } else if (item < O_CH) {
                addr = phiE + (item - O_TE) * N + s;
            } else if (item == O_CH) {
                addr = coshK + s;
            } else if (item == O_SH) {
                addr = sinhK + s;
            } else if (CDW != 0 && item >= O_CC && item <= O_CL) {
                addr = (item == O_CC ? dm.cdwC : item == O_SC ? dm.cdwS : dm.cdwl) + (size_t)k * N + s;
            } else if (item >= O_GCC && item < NITEMS) {
                const int t = item - O_GCC;
                const int blk = t / (2 * MSF * MSF), e = t % (2 * MSF * MSF);
                const int part = e & 1, ab = e >> 1, a = ab / MSF, b = ab % MSF;
                const int row = (blk == 2 ? prev : s) - goff + a * gNB, col = (blk == 1 ? prev : s) - goff + b * gNB;
                if (blk == 0 || prev >= 0) addr = (const double*)G + 2 * ((size_t)col * gld + row) + part;
            }
            pre[q] = addr ? *addr : dflt;
        }
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int t = lane + 64 * q;                  // (i, a): entry u[a][i] = G[c_a, I_i] and v[i][a] = G[I_i, c_a]
            if (t < MSF * nIknown) {
                const int a = t % MSF, i = t / MSF;
                const int Ii = isite[i / MSF] - goff + (i % MSF) * gNB, sa = s - goff + a * gNB;
                pu[q] = G[(size_t)Ii * gld + sa];
                pv[q] = G[(size_t)sa * gld + Ii];
            }
        }
    };

    // developer phase timers (build with -DDQMC_DECIDE_TIMING, run with DQMC_DECIDE_TIMING=1): cycles between TICK marks,
    // accumulated over the launch.  Compiled out by default -- the counters cost ~26 SGPRs in a kernel that is
    // already short of them.
#ifdef DQMC_DECIDE_TIMING
    unsigned long long tk[12], tlast = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) tk[i] = 0;
    const bool timing = (dm.dbg & 8) != 0;
#define TICK(n) do { if (timing) { unsigned long long t_ = __builtin_readcyclecounter(); tk[n] += t_ - tlast; tlast = t_; } } while (0)
    if (timing) tlast = __builtin_readcyclecounter();
#else
#define TICK(n) do { } while (0)
#endif
    int j = 0;
    const int dnow = min(D, N - site);            // delayStepsNow (:3052)
    // A block also ends after `pbudget` PROPOSALS (0: no limit).  Where a block ends does not change the chain (the flush is exact,
    // test_update_slice_delay_steps_invariance), but in a batched launch every chain waits for the slowest one: the number of
    // proposals until D are accepted scatters (68 +- 9 at acceptance 0.5, D = 32: the slowest of 128 chains needs ~ 91), a
    // proposal budget makes the chains of a launch finish together.
    const int budget = dm.pbudget > 0 ? dm.pbudget : N;
    double pre[NIT];                               // wave 0: the prefetched scalar items, one per lane
    cplx pu[SLOTS], pv[SLOTS];                     // wave 0: u = G[c, I], v = G[I, c] for the I known at issue time
#pragma unroll
    for (int q = 0; q < NIT; ++q) pre[q] = 0.0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) { pu[q] = make_double2(0.0, 0.0); pv[q] = make_double2(0.0, 0.0); }
    fetch(pre, pu, pv, site, -1, cur, 0);
    // No global store happens inside the loop: a store followed by a load that may alias makes the compiler wait
    // for the store's round trip (s_waitcnt vmcnt(0)) before every prefetch.  The slice's field lives in LDS.
    {   // eight loads in flight per thread (written as `sphi[t] = phik[t]` the loop compiles to load, s_waitcnt vmcnt(0), ds_write per
        // element: seven dependent trips to memory in front of the first proposal of an O(3) L = 24 launch)
        constexpr int UB = 8;
        const int total = OPDIM * N;
        for (int base = tid; base < total; base += UB * NT) {
            double r[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) r[u] = phik[min(base + u * NT, total - 1)];
#pragma unroll
            for (int u = 0; u < UB; ++u) { const int t = base + u * NT; if (t < total) sphi[t] = r[u]; }
        }
    }
    int cnd_nI = 0;                                // nI the u/v registers were loaded for
    [[maybe_unused]] unsigned long long win_cur = cur;      // PROP != 0: cursor the current candidate's window of uniforms starts at
    // PROP 2 / 3: the Box-Muller stack of NormalDistribution (normaldistribution.h:44-78) -- after a pair has been generated its first
    // value waits for the next get(); reset at the top of updateInSlice (:2433-2435), kept across the launches and repeats of a slice
    [[maybe_unused]] bool nd_has = (PROP >= 2 && !reset_nd) ? (us->nd_has != 0) : false;
    [[maybe_unused]] double nd_cached = (PROP >= 2 && !reset_nd) ? us->nd_cached : 0.0;
    [[maybe_unused]] const double angleDelta = us->pub.angleDelta, scaleDelta = us->pub.scaleDelta;
    int prev_site = -1;                            // site decided in the previous iteration
    bool prev_acc = false, prev_used_uniform = true;

    int it = 0;                                    // proposal counter of this launch: selects the u/v buffer
    while (j < dnow && site < N && it < budget) {
        if (cur + (PROP == 0 ? NPROP + 1 : 2 * UNIW) > avail) { err = DQMC_ERNG; break; }
        const int nI = MSF * j;
        cplx* su = su2 + (it & 1) * MSF * WD;
        cplx* sv = sv2 + (it & 1) * WD * MSF;
        // ---- A (wave 0): land the prefetched scalars and u / v in LDS; patch in what the previous decision
        //      changed.  The u/v buffers alternate, so threads still reading the previous proposal's u are not
        //      overtaken; scand is read right after barrier 1 and rewritten only after barrier 2. ----
        if (wave == FETCH_WAVE) {
#pragma unroll
            for (int q = 0; q < NIT; ++q) scand[lane + 64 * q] = pre[q];
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const int t = lane + 64 * q;
                if (t < MSF * cnd_nI) {
                    const int a = t % MSF, i = t / MSF;
                    su[a * WD + i] = pu[q];
                    sv[i * MSF + a] = pv[q];
                }
            }
            // the site accepted last time joined I after the fetch: its u / v entries are G[c, prev], G[prev, c],
            // which sit in the lanes O_GNP.. / O_GPN.. of `pre`
            const int e = (lane < MSF * MSF) ? lane : 0;
            constexpr int QN = O_GNP / 64, QP = O_GPN / 64;
            static_assert((O_GNP + 2 * MSF * MSF - 1) / 64 == QN && (O_GPN + 2 * MSF * MSF - 1) / 64 == QP, "item block straddles a lane group");
            cplx gnp = make_double2(__shfl(pre[QN], (O_GNP + 2 * e) & 63, 64), __shfl(pre[QN], (O_GNP + 2 * e + 1) & 63, 64));
            cplx gpn = make_double2(__shfl(pre[QP], (O_GPN + 2 * e) & 63, 64), __shfl(pre[QP], (O_GPN + 2 * e + 1) & 63, 64));
            if (prev_acc && lane < MSF * MSF) {
                const int a = lane / MSF, b = lane % MSF;
                su[a * WD + (nI - MSF + b)] = gnp;          // u[a][i] = G[c_a, I_i], I_i = prev + b N
                sv[(nI - MSF + a) * MSF + b] = gpn;         // v[i][b] = G[I_i, c_b], I_i = prev + a N
            }
        }
        TICK(0);
        __syncthreads();                              // barrier 1 of 2: scalars, u, v (and last proposal's W update) visible
        TICK(1);
        // ---- C: start the loads of the NEXT candidate now; they complete while this decision is computed ----
        // not behind the last proposal of the budget: that candidate belongs to the next launch, and with a window copy (winP > 0)
        // its entries lie outside the (MSF P)^2 window this launch may read
        const bool have_next = (site + 1 < N) && (it + 1 < budget);
        if (have_next) fetch(pre, pu, pv, site + 1, site, PROP == 0 ? cur + NPROP : cur, nI);
        [[maybe_unused]] const unsigned long long fetch_cur_next = cur;     // PROP != 0: where the NEXT candidate's window starts
        // ---- the waves split the work between the two barriers: wave 0 does the scalar Metropolis arithmetic of this
        //      proposal (D) and hands delta, exp(-dS), the acceptance uniform and G[c,c] to the others through LDS; waves
        //      1-3 meanwhile form p and q (E).  Neither waits for the other before barrier 2. ----
        // uniforms: skip the one the previous decision consumed for its acceptance test
        const int uoff = (prev_site >= 0 && prev_used_uniform) ? 1 : 0;
        double newphi[OPDIM], coshN = 0.0, sinhN = 0.0;
#pragma unroll
        for (int d = 0; d < OPDIM; ++d) newphi[d] = 0.0;
        if constexpr (PROP == 0) cur += NPROP;             // (PROP != 0: the count comes back from wave 0 behind barrier 2)
        double lnew = 0.0, cCn = 1.0, sCn = 0.0;          // CDW 2: the proposed l and its cosh / sinh terms
        TICK(2);
        if (wave >= E_FIRST) {
            // ---- E: p = W v and, speculatively (needed only on acceptance), q = u W.  Work item = one ROW of p (all MSF entries
            //      from one pass over row i of W) or one COLUMN of q, split over a quad of lanes; items [0, nI) are p(i, :), items
            //      [nI, 2 nI) are q(:, i).  Two independent accumulator sets per lane (even / odd trips) keep two rounds of LDS reads
            //      in flight: the dependent fma chain of a single accumulator left the LDS latency of every trip exposed. ----
            const int nitems = 2 * nI;
            for (int t = tid - 64 * E_FIRST; t < 4 * nitems; t += 64 * NWE) {
                const int item = t >> 2, part = t & 3;
                cplx acc0[MSF], acc1[MSF];
#pragma unroll
                for (int b = 0; b < MSF; ++b) { acc0[b] = make_double2(0.0, 0.0); acc1[b] = make_double2(0.0, 0.0); }
                const bool isp = item < nI;
                const int i = isp ? item : item - nI;
                // p: W[i][i2] v[i2][b];  q: u[a][i2] W[i2][i]
                const cplx* wp = isp ? W + i * WS : W + i;
                const int wstep = isp ? 1 : WS;
                const cplx* xp = isp ? sv : su;
                const int xstep = isp ? MSF : 1, xsel = isp ? 1 : WD;        // operand (i2, b): xp[i2 * xstep + b * xsel]
                int i2 = part;
                for (; i2 + 4 < nI; i2 += 8) {
                    const cplx w0 = wp[i2 * wstep], w1 = wp[(i2 + 4) * wstep];
#pragma unroll
                    for (int b = 0; b < MSF; ++b) {
                        const cplx x0 = xp[i2 * xstep + b * xsel], x1 = xp[(i2 + 4) * xstep + b * xsel];
                        acc0[b] = u_cfma(w0, x0, acc0[b]);
                        acc1[b] = u_cfma(w1, x1, acc1[b]);
                    }
                }
                if (i2 < nI) {
                    const cplx w0 = wp[i2 * wstep];
#pragma unroll
                    for (int b = 0; b < MSF; ++b) acc0[b] = u_cfma(w0, xp[i2 * xstep + b * xsel], acc0[b]);
                }
#pragma unroll
                for (int b = 0; b < MSF; ++b) {
                    cplx acc = make_double2(acc0[b].x + acc1[b].x, acc0[b].y + acc1[b].y);
                    acc.x = u_dpp_add<0xB1, 0xf>(acc.x); acc.y = u_dpp_add<0xB1, 0xf>(acc.y);     // quad sum
                    acc.x = u_dpp_add<0x4E, 0xf>(acc.x); acc.y = u_dpp_add<0x4E, 0xf>(acc.y);
                    if (part == 0) { if (isp) sp[i * MSF + b] = acc; else sq[b * WD + i] = acc; }
                }
            }
        } else if (wave == 0) {
            // ---- D (wave 0): the candidate's scalars (broadcast LDS reads), bosonic action (deltaSPhi, :4186-4239)
            //      and delta (get_delta_forsite, :3179-3289) ----
            int nbr[4];
            neighbours(site, nbr);
            double oldphi[OPDIM], snb[OPDIM], tnb[OPDIM];
            [[maybe_unused]] int consumed = 0;           // PROP != 0: uniforms this proposal drew
            [[maybe_unused]] bool valid = true;          //            changed != NONE
            [[maybe_unused]] double uacc_peek = 0.5;
            if constexpr (PROP != 0) {
                int wpos = (int)(cur - win_cur);
                auto next_uniform = [&]() -> double {
                    const int wp = wpos + consumed;
                    double v;
                    if (wp < UNIW) v = scand[O_UNI + wp];
                    else { const unsigned long long idx = cur + consumed; v = idx < avail ? uni[idx] : 0.25; }
                    consumed += 1;
                    return v;
                };
                double ov[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) ov[d] = sphi[d * N + site];
                [[maybe_unused]] auto nd_get = [&](double sigma, double mean) -> double {     // NormalDistribution::get
#pragma clang fp contract(off)
                    double var;
                    if (!nd_has) {
                        double v1, v2, rsq;
                        int guard = 0;
                        do {
                            const double u1 = next_uniform(), u2 = next_uniform();
                            v1 = 2.0 * u1 - 1.0; v2 = 2.0 * u2 - 1.0;
                            rsq = v1 * v1 + v2 * v2;
                        } while ((rsq >= 1.0 || rsq == 0.0) && ++guard < 64);
                        const double fac = sqrt(-2.0 * log(rsq) / rsq);
                        nd_cached = v1 * fac;            // pushed first, handed out second
                        var = v2 * fac;
                        nd_has = true;
                    } else { var = nd_cached; nd_has = false; }
                    const double t = sigma * var;
                    return mean + t;
                };
                double nv[3] = {ov[0], ov[1], ov[2]};
                if constexpr (PROP == 1) {
#pragma clang fp contract(off)
                    const double a0 = next_uniform(), a1 = next_uniform();
                    const double span = 1.0 - angleDelta;
                    const double ct = a0 * span + angleDelta;
                    const double ph = a1 * 2.0 * M_PI;
                    double r, un[3];
                    rot3_unit(ov, ct, ph, r, un);
                    nv[0] = un[0] * r; nv[1] = un[1] * r; nv[2] = un[2] * r;
                } else if constexpr (PROP == 2) {
#pragma clang fp contract(off)
                    const double r3 = pow(ov[0] * ov[0] + ov[1] * ov[1] + ov[2] * ov[2], 1.5);
                    const double new_r3 = nd_get(scaleDelta, r3);
                    if (new_r3 <= 0) valid = false;
                    else {
                        const double q = new_r3 / r3;
                        const double scale = pow(q, 1.0 / 3.0);
                        nv[0] = ov[0] * scale; nv[1] = ov[1] * scale; nv[2] = ov[2] * scale;
                    }
                } else {
#pragma clang fp contract(off)
                    const double rr = sqrt(ov[0] * ov[0] + ov[1] * ov[1] + ov[2] * ov[2]);
                    const double r3 = pow(rr, 3.0);
                    const double new_r3 = nd_get(scaleDelta, r3);
                    if (new_r3 <= 0) valid = false;
                    else {
                        const double a0 = next_uniform(), a1 = next_uniform();
                        const double span = 1.0 - angleDelta;
                        const double ct = a0 * span + angleDelta;
                        const double ph = a1 * 2.0 * M_PI;
                        double r, un[3];
                        rot3_unit(ov, ct, ph, r, un);
                        const double new_r = pow(new_r3, 1.0 / 3.0);
                        nv[0] = un[0] * new_r; nv[1] = un[1] * new_r; nv[2] = un[2] * new_r;
                    }
                }
#pragma unroll
                for (int d = 0; d < OPDIM; ++d) newphi[d] = nv[d < 3 ? d : 0];
                {   // the uniform an acceptance test would draw next (not consumed here)
                    const int wp = wpos + consumed;
                    if (wp < UNIW) uacc_peek = scand[O_UNI + wp];
                    else { const unsigned long long idx = cur + consumed; uacc_peek = idx < avail ? uni[idx] : 0.5; }
                }
            }
#pragma unroll
            for (int d = 0; d < OPDIM; ++d) {
                oldphi[d] = sphi[d * N + site];
                double low = -phiDelta, high = phiDelta;
                if constexpr (PROP == 0) newphi[d] = (CDW == 2) ? oldphi[d] : propose_component(oldphi[d], low, high, scand[O_UNI + uoff + d]);
                // XPLUS, XMINUS, YPLUS, YMINUS in the order of the reference's neighbour loop (:4208-4214)
                snb[d] = ((0.0 + sphi[d * N + nbr[0]]) + sphi[d * N + nbr[1]]) + sphi[d * N + nbr[2]] + sphi[d * N + nbr[3]];
                tnb[d] = scand[O_TL + d] + scand[O_TE + d];
            }
            const double coshO = scand[O_CH], sinhO = scand[O_SH];
            double dsphi = 0.0;
            if constexpr (CDW != 2) {
                double oldSq = 0.0, newSq = 0.0;
#pragma unroll
                for (int d = 0; d < OPDIM; ++d) { oldSq += oldphi[d] * oldphi[d]; newSq += newphi[d] * newphi[d]; }
                double phiSqDiff = newSq - oldSq;
                if (dm.phi2bosons) {
                    dsphi = dm.dtau * 0.5 * dm.r * phiSqDiff;
                } else {
                    double phiPow4Diff = newSq * newSq - oldSq * oldSq;
                    double dotTime = 0.0, dotSpace = 0.0;
#pragma unroll
                    for (int d = 0; d < OPDIM; ++d) {
                        double diff = newphi[d] - oldphi[d];
                        dotTime += tnb[d] * diff;
                        dotSpace += snb[d] * diff;
                    }
                    double delta1 = (1.0 / (dm.c * dm.c * dm.dtau)) * (phiSqDiff - dotTime);
                    double delta2 = 0.5 * dm.dtau * (4.0 * phiSqDiff - 2.0 * dotSpace);
                    double delta3 = dm.dtau * (0.5 * dm.r * phiSqDiff + 0.25 * dm.u * phiPow4Diff);
                    dsphi = delta1 + delta2 + delta3;
                }
            }
            double probSPhi;
            // cdw terms of the site as it is (cC, sC) and as proposed (cCn, sCn)
            double cCo = 1.0, sCo = 0.0;
            if constexpr (CDW != 0) { cCo = scand[O_CC]; sCo = scand[O_SC]; cCn = cCo; sCn = sCo; }
            bool nullp = false;
            if constexpr (CDW == 2) {
                const double r01 = scand[O_UNI + uoff], lold = scand[O_CL];
                lnew = (r01 <= 0.25) ? 2.0 : (r01 <= 0.5) ? -2.0 : (r01 <= 0.75) ? 1.0 : -1.0;
                const int an = (fabs(lnew) > 1.5) ? 1 : 0, ao = (fabs(lold) > 1.5) ? 1 : 0;
                cCn = dm.cdw_cosh[an];
                sCn = (lnew < 0.0) ? -dm.cdw_sinh[an] : dm.cdw_sinh[an];
                probSPhi = dm.cdw_gamma[an] / dm.cdw_gamma[ao];       // prob_cdwl; probSPhi itself is 1 for a CDWL proposal
                // The proposal drew the value the site already has: delta = 0 and the probability is 1 in exact arithmetic -- a
                // uniform is drawn, nothing changes (the reference's floating point lands on 1 or 1 + 2^-52 there and skips the uniform
                // in the second case)
                nullp = (lnew == lold);
                coshN = coshO; sinhN = sinhO;
            } else {
                double nn = 0.0;
#pragma unroll
                for (int d = 0; d < OPDIM; ++d) nn += newphi[d] * newphi[d];
                double nrm = sqrt(nn);
                double arg = dm.lambda * dm.dtau * nrm;
                // the wave evaluates ONE exp sequence: lane 0 on -dS, the other lanes on arg
                double ex = exp(lane == 0 ? -dsphi : arg);
                int lo = __builtin_amdgcn_readlane(__double2loint(ex), 0), hi = __builtin_amdgcn_readlane(__double2hiint(ex), 0);
                probSPhi = __hiloint2double(hi, lo);
                lo = __builtin_amdgcn_readlane(__double2loint(ex), 1); hi = __builtin_amdgcn_readlane(__double2hiint(ex), 1);
                double ea = __hiloint2double(hi, lo), iea = 1.0 / ea;
                coshN = 0.5 * (ea + iea);
                // sinh(a)/|phi|: for small a the difference e^a - e^-a cancels, use the series there
                double sh = (arg > 0.25) ? 0.5 * (ea - iea)
                                         : arg * (1.0 + arg * arg * (1.0 / 6.0 + arg * arg * (1.0 / 120.0 + arg * arg *
                                                  (1.0 / 5040.0 + arg * arg * (1.0 / 362880.0 + arg * arg * (1.0 / 39916800.0))))));
                sinhN = sh / nrm;
            }
            cplx evOld[MSF][MSF], emvNew[MSF][MSF];
            if constexpr (CDW == 0) {
                ev_matrix<MSF>(evOld, +1.0, oldphi, OPDIM, coshO, coshO, sinhO);
                ev_matrix<MSF>(emvNew, -1.0, newphi, OPDIM, coshN, coshN, sinhN);
            } else {
                ev_matrix<MSF>(evOld, +1.0, oldphi, OPDIM, coshO * cCo - sCo, coshO * cCo + sCo, sinhO * cCo);
                ev_matrix<MSF>(emvNew, -1.0, newphi, OPDIM, coshN * cCn + sCn, coshN * cCn - sCn, sinhN * cCn);
            }
            if (tid < MSF * MSF) {             // lane (a, b) publishes delta[a][b] and G[c,c][a][b]
                cplx dsel = make_double2(0.0, 0.0);
                if constexpr (MSF == 4) {
                    // O(3): lane (a, b) forms ITS entry only -- row a of e^{-dtau V'} and column b of e^{+dtau V} picked by selects, the four
                    // terms in the same order as before (round 3 had every lane form all 16 entries and keep one: 256 dependent-issue
                    // fp64 FMAs on the scalar leg of every proposal)
                    const int la = tid >> 2, lb = tid & 3;
                    cplx acc = make_double2(la == lb ? -1.0 : 0.0, 0.0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // (bit-mask picks: written as ?: chains the compiler turns them into a dynamically indexed array in scratch)
                        const cplx r = u_pick4(la, emvNew[0][q], emvNew[1][q], emvNew[2][q], emvNew[3][q]);
                        const cplx c = u_pick4(lb, evOld[q][0], evOld[q][1], evOld[q][2], evOld[q][3]);
                        acc = u_cfma(r, c, acc);
                    }
                    dsel = acc;
                } else {
#pragma unroll
                for (int a = 0; a < MSF; ++a)
#pragma unroll
                    for (int b = 0; b < MSF; ++b) {
                        cplx acc = make_double2(a == b ? -1.0 : 0.0, 0.0);
#pragma unroll
                        for (int q = 0; q < MSF; ++q) acc = u_cfma(emvNew[a][q], evOld[q][b], acc);
                        if (tid == a * MSF + b) dsel = acc;
                    }
                }
                *(cplx*)&sdec[2 * tid] = dsel;
                *(cplx*)&sdec[2 * MSF * MSF + 2 * tid] = *(const cplx*)&scand[O_GCC + 2 * tid];
            }
            if (tid == 0) {
                sdec[4 * MSF * MSF] = probSPhi;
                if constexpr (PROP == 0) sdec[4 * MSF * MSF + 1] = scand[O_UNI + uoff + NPROP];
                else { sdec[4 * MSF * MSF + 1] = uacc_peek; sdec[4 * MSF * MSF + 2] = valid ? 0.0 : 2.0; sdec[4 * MSF * MSF + 3] = (double)consumed; }
                if constexpr (CDW == 2) { sdec[4 * MSF * MSF + 2] = nullp ? 1.0 : 0.0; sdec[4 * MSF * MSF + 3] = lnew; }
            }
        }
        TICK(4);
        __syncthreads();                              // barrier 2 of 2: p, q visible
        TICK(5);
        // ---- every thread picks up what wave 0 published ----
        const double probSPhi = sdec[4 * MSF * MSF], uacc = sdec[4 * MSF * MSF + 1];
        cplx delta[MSF == 2 ? 2 : 1][MSF == 2 ? 2 : 1], Minv[MSF == 2 ? 2 : 1][MSF == 2 ? 2 : 1];
        cplx det;
        if constexpr (MSF == 2) {
            cplx Gcc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    delta[a][b] = *(const cplx*)&sdec[2 * (a * 2 + b)];
                    Gcc[a][b] = *(const cplx*)&sdec[2 * 4 + 2 * (a * 2 + b)];
                }
            // ---- G: S = Gcc + u p ----
            cplx S[2][2];
            // 16 lanes per entry (a, b) = (lane >> 5, (lane >> 4) & 1); sum within the DPP row, then 8 readlanes
            const int e_a = lane >> 5, e_b = (lane >> 4) & 1, l16 = lane & 15;
            cplx part = make_double2(0.0, 0.0);
            for (int i = l16; i < nI; i += 16) part = u_cfma(su[e_a * WD + i], sp[i * MSF + e_b], part);
            part.x = u_dpp_add<0x111, 0xf>(part.x); part.y = u_dpp_add<0x111, 0xf>(part.y);   // row_shr:1
            part.x = u_dpp_add<0x112, 0xf>(part.x); part.y = u_dpp_add<0x112, 0xf>(part.y);   // row_shr:2
            part.x = u_dpp_add<0x114, 0xf>(part.x); part.y = u_dpp_add<0x114, 0xf>(part.y);   // row_shr:4
            part.x = u_dpp_add<0x118, 0xf>(part.x); part.y = u_dpp_add<0x118, 0xf>(part.y);   // row_shr:8 -> lane 15 of the row
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int src = (a * 2 + b) * 16 + 15;
                    double re = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(part.x), src),
                                                 __builtin_amdgcn_readlane(__double2loint(part.x), src));
                    double im = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(part.y), src),
                                                 __builtin_amdgcn_readlane(__double2loint(part.y), src));
                    S[a][b] = make_double2(Gcc[a][b].x + re, Gcc[a][b].y + im);
                }
            TICK(6);
            // ---- M' = 1 + (1 - S) delta ; det ; acceptance ----
            cplx Mj[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    cplx acc = make_double2((a == b ? 1.0 : 0.0) + delta[a][b].x, delta[a][b].y);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc = u_cfma(make_double2(-S[a][q].x, -S[a][q].y), delta[q][b], acc);
                    Mj[a][b] = acc;
                }
            det = small_det_inv<2>(Mj, Minv);
        } else {
            // ---- O(3): the 4 x 4 complex algebra of a decision is spread over the lanes of the wave instead of being done in full by
            //      every thread (round 2: 16 wave-wide reductions for S, a Gauss-Jordan inverse with run-time pivot selects in each
            //      thread -- about a thousand fp64 instructions and 326 registers; 5.4 us per proposal at n_g = 2304).  Lane (e, sub):
            //      entry e = (a, b) of a 4 x 4 matrix, sub = one of the four terms of its sum; the matrices hop through a wave-private
            //      LDS scratch (LDS operations of one wave complete in order: no barrier).  Inverse by cofactors (M' = 1 + (1 - S)
            //      delta is a well-conditioned 4 x 4 matrix; the reference's LAPACK inverse agrees to rounding). ----
            cplx* aS = salg + (tid >> 6) * 80;
            cplx* aM = aS + 16; cplx* aC = aS + 32; cplx* aI = aS + 48;
            const int e = lane >> 2, sub = lane & 3, ea = e >> 2, eb = e & 3;
            auto quadsum = [&](cplx v) {
                v.x = u_dpp_add<0xB1, 0xf>(v.x); v.y = u_dpp_add<0xB1, 0xf>(v.y);
                v.x = u_dpp_add<0x4E, 0xf>(v.x); v.y = u_dpp_add<0x4E, 0xf>(v.y);
                return v;
            };
            {   // S = Gcc + u p
                cplx part = make_double2(0.0, 0.0);
                for (int i = sub; i < nI; i += 4) part = u_cfma(su[ea * WD + i], sp[i * MSF + eb], part);
                part = quadsum(part);
                const cplx g = *(const cplx*)&sdec[2 * MSF * MSF + 2 * e];
                if (sub == 0) aS[e] = make_double2(g.x + part.x, g.y + part.y);
            }
            TICK(6);
            {   // M' = 1 + delta - S delta: lane sub carries the term q = sub
                const cplx t = quadsum(u_cmul(aS[ea * 4 + sub], *(const cplx*)&sdec[2 * (sub * 4 + eb)]));
                const cplx d = *(const cplx*)&sdec[2 * e];
                if (sub == 0) aM[e] = make_double2((ea == eb ? 1.0 : 0.0) + d.x - t.x, d.y - t.y);
            }
            {   // cofactor of entry (ea, eb): signed 3 x 3 minor
                const int r0 = (ea == 0) ? 1 : 0, r1 = (ea <= 1) ? 2 : 1, r2 = (ea == 3) ? 2 : 3;
                const int c0 = (eb == 0) ? 1 : 0, c1 = (eb <= 1) ? 2 : 1, c2 = (eb == 3) ? 2 : 3;
                const cplx m00 = aM[r0 * 4 + c0], m01 = aM[r0 * 4 + c1], m02 = aM[r0 * 4 + c2];
                const cplx m10 = aM[r1 * 4 + c0], m11 = aM[r1 * 4 + c1], m12 = aM[r1 * 4 + c2];
                const cplx m20 = aM[r2 * 4 + c0], m21 = aM[r2 * 4 + c1], m22 = aM[r2 * 4 + c2];
                const cplx k0 = u_csub(u_cmul(m11, m22), u_cmul(m12, m21));
                const cplx k1 = u_csub(u_cmul(m10, m22), u_cmul(m12, m20));
                const cplx k2 = u_csub(u_cmul(m10, m21), u_cmul(m11, m20));
                cplx d3 = u_csub(u_cmul(m00, k0), u_cmul(m01, k1));
                d3 = u_cfma(m02, k2, d3);
                const double sg = ((ea + eb) & 1) ? -1.0 : 1.0;
                if (sub == 0) aC[e] = make_double2(sg * d3.x, sg * d3.y);
            }
            det = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < 4; ++q) det = u_cfma(aM[q], aC[q], det);          // expansion along row 0
            {
                const double dn = det.x * det.x + det.y * det.y;
                const cplx idet = make_double2(det.x / dn, -det.y / dn);
                if (sub == 0) aI[e] = u_cmul(aC[eb * 4 + ea], idet);              // M'^-1 = adj(M') / det
            }
        }
        double probSFermion = (OPDIM == 3) ? det.x : (det.x * det.x + det.y * det.y);
        double prob = probSPhi * probSFermion;
        bool accept = prob > 1.0;
        bool used_uniform = false;
        if constexpr (CDW == 2) {
            if (sdec[4 * MSF * MSF + 2] != 0.0) { prob = 1.0; accept = false; }      // null proposal: the uniform is drawn, the state stays
        }
        [[maybe_unused]] bool dropped = false;
        if constexpr (PROP != 0) {
            cur += (unsigned long long)sdec[4 * MSF * MSF + 3];                        // what the proposal drew
            dropped = sdec[4 * MSF * MSF + 2] != 0.0;                                  // changed == NONE (:3063): no acceptance draw at all
            if (dropped) accept = false;
        }
        if (!accept && !dropped) { accept = uacc < prob; cur += 1; used_uniform = true; }   // rand01 drawn only if prob <= 1 (:3113)
        if constexpr (CDW == 2) {
            if (sdec[4 * MSF * MSF + 2] != 0.0) accept = false;
        }
        TICK(7);
        if (accept) {
            acc_count += 1;
            if (tid == 0) {
                if constexpr (CDW == 2) {
                    sacc[j][0] = lnew; sacc[j][1] = cCn; sacc[j][2] = sCn;
                } else {
#pragma unroll
                    for (int d = 0; d < OPDIM; ++d) { sphi[d * N + site] = newphi[d]; sacc[j][d] = newphi[d]; }
                    sacc[j][OPDIM] = coshN;
                    sacc[j][OPDIM + 1] = sinhN;
                }
                isite[j] = site;
            }
            // F = delta M'^-1
            cplx F[MSF][MSF];
            if constexpr (MSF == 2) {
#pragma unroll
                for (int a = 0; a < MSF; ++a)
#pragma unroll
                    for (int b = 0; b < MSF; ++b) {
                        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                        for (int q = 0; q < MSF; ++q) acc = u_cfma(delta[a][q], Minv[q][b], acc);
                        F[a][b] = acc;
                    }
            } else {
                cplx* aI = salg + (tid >> 6) * 80 + 48;
                cplx* aF = salg + (tid >> 6) * 80 + 64;
                const int e = lane >> 2, sub = lane & 3, ea = e >> 2, eb = e & 3;
                cplx t = u_cmul(*(const cplx*)&sdec[2 * (ea * 4 + sub)], aI[sub * 4 + eb]);
                t.x = u_dpp_add<0xB1, 0xf>(t.x); t.y = u_dpp_add<0xB1, 0xf>(t.y);
                t.x = u_dpp_add<0x4E, 0xf>(t.x); t.y = u_dpp_add<0x4E, 0xf>(t.y);
                if (sub == 0) aF[e] = t;
#pragma unroll
                for (int a = 0; a < MSF; ++a)
#pragma unroll
                    for (int b = 0; b < MSF; ++b) F[a][b] = aF[a * MSF + b];
            }
            // block bordering of W straight from p and q in LDS (pF is formed on the fly: no staging, no barrier):
            //   W11 += (p F) q ;  W12 = p F ;  W21 = F q ;  W22 = F
            // A lane owns column i2 of W11 (its q entries stay in registers), a wave walks the rows i = w, w + NWU, ...: no
            // integer division by nI, the row's p entries are broadcast reads, four rows in flight per trip.
            if (wave >= U_FIRST && lane < nI) {
                cplx qc[MSF];
#pragma unroll
                for (int b = 0; b < MSF; ++b) qc[b] = sq[b * WD + lane];
                const int wv = wave - U_FIRST;
                for (int i = wv; i < nI; i += 4 * NWU) {
                    cplx wold[4], pf[4][MSF];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ir = min(i + NWU * r, nI - 1);
                        wold[r] = W[ir * WS + lane];
                        cplx pi[MSF];
#pragma unroll
                        for (int q = 0; q < MSF; ++q) pi[q] = sp[ir * MSF + q];
#pragma unroll
                        for (int b = 0; b < MSF; ++b) {
                            cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                            for (int q = 0; q < MSF; ++q) acc = u_cfma(pi[q], F[q][b], acc);
                            pf[r][b] = acc;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cplx acc = wold[r];
#pragma unroll
                        for (int b = 0; b < MSF; ++b) acc = u_cfma(pf[r][b], qc[b], acc);
                        if (i + NWU * r < nI) W[(i + NWU * r) * WS + lane] = acc;
                    }
                }
            }
            for (int i = tid; i < nI; i += 64 * NW) {          // nI <= 64: wave 0
                cplx pi[MSF];
#pragma unroll
                for (int q = 0; q < MSF; ++q) pi[q] = sp[i * MSF + q];
#pragma unroll
                for (int b = 0; b < MSF; ++b) {
                    cplx pf = make_double2(0.0, 0.0), fq = make_double2(0.0, 0.0);
#pragma unroll
                    for (int q = 0; q < MSF; ++q) {
                        pf = u_cfma(pi[q], F[q][b], pf);
                        fq = u_cfma(F[b][q], sq[q * WD + i], fq);
                    }
                    W[i * WS + (nI + b)] = pf;
                    W[(nI + b) * WS + i] = fq;
                }
            }
            if (tid == 0) {
#pragma unroll
                for (int a = 0; a < MSF; ++a)
#pragma unroll
                    for (int b = 0; b < MSF; ++b) W[(nI + a) * WS + (nI + b)] = F[a][b];
            }
            j += 1;
        }
        TICK(8);
        // hand over to the next candidate (no barrier here: the next proposal writes the OTHER u/v buffer, and
        // everything else it touches before its first barrier is thread private)
        prev_site = site;
        prev_acc = accept;
        prev_used_uniform = used_uniform;
        cnd_nI = nI;                               // what the u/v registers were fetched for; the accepted site is patched in
        if constexpr (PROP != 0) win_cur = fetch_cur_next;
        site += 1;
        it += 1;
        TICK(9);
    }
    __syncthreads();                               // W complete before it is published
    // the accepted field values and their cosh / sinh terms (updateCoshSinhTerms, :3128-3140)
    if (tid < j) {
        const int st = isite[tid];
        if constexpr (CDW == 2) {
            dm.cdwl[(size_t)k * N + st] = sacc[tid][0];
            dm.cdwC[(size_t)k * N + st] = sacc[tid][1];
            dm.cdwS[(size_t)k * N + st] = sacc[tid][2];
        } else {
#pragma unroll
            for (int d = 0; d < OPDIM; ++d) phik[d * N + st] = sacc[tid][d];
            dm.coshT[(size_t)k * N + st] = sacc[tid][OPDIM];
            dm.sinhT[(size_t)k * N + st] = sacc[tid][OPDIM + 1];
        }
    }

    // ---- publish block result ----
    const int nI = MSF * j;
    for (int t = tid; t < nI * nI; t += NT) {
        int i = t / nI, i2 = t - i * nI;
        Wout[(size_t)i2 * WD + i] = W[i * WS + i2];        // column-major, ld = WD
    }
    if (tid == 0) {
        us->site_cursor = site;
        us->acc_count = acc_count;
        us->block_j = j;
        if (j > 0) { us->blocks_nonempty += 1; us->updates_accepted += (unsigned long long)j; }
        for (int l = 0; l < j; ++l) us->block_sites[l] = isite[l];
        us->pub.rng_consumed = cur;
        us->pub.error = err;
        if constexpr (PROP >= 2) { us->nd_has = nd_has ? 1 : 0; us->nd_cached = nd_cached; }
        int sdone = 0;
        if (site >= N && err == 0) {
            sdone = 1;
            double accratio = (double)acc_count / (double)N;            // :3173
            if (CDW != 2) us->pub.lastAccRatio = accratio;              // the cdwl pass's ratio is discarded (:2476-2477)
            if (thermal && CDW != 2) {
                // RunningAverage::addValue (RunningAverage.h:57-68), sampleSize = 100, of the average that belongs to the kind of move
                // this pass made (accRatioLocal_box_RA / _rotate_RA / _scale_RA, :3322-3329)
                const int sampleSize = 100;
                const int what = adapt_what & 3;
                double* rap = what == 0 ? &us->pub.ra_runningAverage : what == 1 ? &us->pub.rot_runningAverage : &us->pub.scl_runningAverage;
                double* vals = what == 0 ? us->pub.ra_values : what == 1 ? us->pub.rot_values : us->pub.scl_values;
                int32_t* addp = what == 0 ? &us->pub.ra_samplesAdded : what == 1 ? &us->pub.rot_samplesAdded : &us->pub.scl_samplesAdded;
                int32_t* headp = what == 0 ? &us->pub.ra_head : what == 1 ? &us->pub.rot_head : &us->pub.scl_head;
                double ra = *rap;
                int added = *addp, head = *headp;
                if (added < sampleSize) {
                    vals[added] = accratio;
                    ra += accratio / sampleSize;
                } else {
                    ra -= vals[head] / sampleSize;
                    vals[head] = accratio;
                    head = (head + 1) % sampleSize;
                    ra += accratio / sampleSize;
                }
                added += 1;
                *rap = ra; *addp = added; *headp = head;
                if (added % sampleSize == 0) {                          // :3331-3375
                    const double tgt = us->pub.targetAccRatio;
                    if (what == 0) {
                        double pd = us->pub.phiDelta;
                        if (ra < tgt) pd *= 0.95;
                        else if (ra > tgt) pd *= 1.05;
                        us->pub.phiDelta = pd;
                    } else if (what == 1) {
                        // angleDelta = minimal cos(theta): bisection between the current bounds, MinAngleDelta = -1, MaxAngleDelta = 1
                        double ad = us->pub.angleDelta;
                        if (ra < tgt && ad < 1.0) { us->pub.curminAngleDelta = ad; ad += (us->pub.curmaxAngleDelta - ad) / 2; }
                        else if (ra > tgt && ad > -1.0) { us->pub.curmaxAngleDelta = ad; ad -= (ad - us->pub.curminAngleDelta) / 2; }
                        us->pub.angleDelta = ad;
                    } else if (adapt_what & 4) {                        // adaptScaleVariance; both branches test ra > target, as the reference does
                        double sd = us->pub.scaleDelta;
                        if (ra > tgt && sd < 1.0) { us->pub.curminScaleDelta = sd; sd += (us->pub.curmaxScaleDelta - sd) / 2; }
                        else if (ra > tgt && sd > 0.0) { us->pub.curmaxScaleDelta = sd; sd -= (sd - us->pub.curminScaleDelta) / 2; }
                        us->pub.scaleDelta = sd;
                    }
                }
            }
        }
        us->slice_done = sdone;
#ifdef DQMC_DECIDE_TIMING
        if (timing) {
            for (int i = 0; i < 12; ++i) us->dbg_cycles[i] += tk[i];
            us->dbg_cycles[12] += 1;
        }
#endif
    }
#undef TICK
}

template<int O, int C, int P> static const void* decide_kernel(bool wide) {
    if constexpr (O != 3) { if (wide) return (const void*)k_update_decide<O, C, P, 512>; }
    return (const void*)k_update_decide<O, C, P, 256>;
}

void launch_update_decide(const Launch& lc, const DevModel* /*dm*/, const DevModel& hm, DevUpdateState* us,
                          const double* uniforms, const cplx* G, cplx* W, int k, int first, int thermal, int cdw_pass,
                          const cplx* Gwin, int winP, int proposal, int adapt_what, int reset_nd) {
    const int WD = hm.MSF * hm.D;
    size_t lds = ((size_t)WD * (WD + 1) + 6 * (size_t)hm.MSF * WD) * sizeof(cplx) + (size_t)hm.opdim * hm.N * sizeof(double);
    const int cdw = hm.cdw_on ? (cdw_pass ? 2 : 1) : 0;
    const int prop = (cdw == 2) ? 0 : proposal;                  // the cdwl pass has its own one-uniform proposal
    const void* f = nullptr;
    int slot = -1;
    // small batches leave most of the chip idle: the 512-thread shape (see k_update_decide) is the faster one there
    const bool wide = hm.opdim != 3 && (hm.decide_nt == 512 || (hm.decide_nt == 0 && lc.nb <= DQMC_DECIDE_WIDE_MAX_CHAINS));
#define DECIDE_CASE(O, C, P) if (hm.opdim == O && cdw == C && prop == P) { \
        f = decide_kernel<O, C, P>(wide); slot = wide ? 36 + (O - 1) * 3 + C : ((O - 1) * 3 + C) * 4 + P; }
    DECIDE_CASE(1, 0, 0) DECIDE_CASE(1, 1, 0) DECIDE_CASE(1, 2, 0) DECIDE_CASE(2, 0, 0) DECIDE_CASE(2, 1, 0) DECIDE_CASE(2, 2, 0)
    DECIDE_CASE(3, 0, 0) DECIDE_CASE(3, 1, 0) DECIDE_CASE(3, 2, 0)
    DECIDE_CASE(3, 0, 1) DECIDE_CASE(3, 0, 2) DECIDE_CASE(3, 0, 3) DECIDE_CASE(3, 1, 1) DECIDE_CASE(3, 1, 2) DECIDE_CASE(3, 1, 3)
#undef DECIDE_CASE
    if (!f) return;            // dqmc_update_slice_ex rejects the combination before it gets here
    if (lds > 48 * 1024) {     // deep delay blocks: raise the dynamic LDS limit of the instantiation to what is needed
        // the attribute belongs to (function, device); several contexts / host threads may get here at once
        static std::mutex mu;
        static size_t raised_tab[64][42] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        size_t& raised = raised_tab[dev & 63][slot];
        if (lds > raised) {
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) raised = lds;
            else (void)hipGetLastError();      // the launch below then reports the problem
        }
    }
    void* args[] = {(void*)&hm, (void*)&us, (void*)&uniforms, (void*)&G, (void*)&W, (void*)&k, (void*)&first, (void*)&thermal, (void*)&lc.cs,
                    (void*)&Gwin, (void*)&winP, (void*)&adapt_what, (void*)&reset_nd};
    (void)hipLaunchKernel(f, dim3(1, 1, lc.nb), dim3(slot >= 36 ? 512 : 256), args, lds, lc.st);
}

// X[:, i'] = sum_i G[:, I_i] W[i, i']   (n_g x nI8, ld n_g);   GrT[:, i] = (G[I_i, :] - E)^T   (n_g x nI8, ld n_g)
// nI8 = nI rounded up to a multiple of 8, the padding columns are ZERO: the flush kernel runs its k loop in steps of 8 without
// a single bounds check.  One workgroup per 32 rows of X / of GrT.
//   * GrT: the rows G[I_i, :] of a column-major G are a strided gather by nature -- but the accepted sites of a block are nearly
//     consecutive (every proposal of a window of <= pbudget sites, about half of them accepted), so for one column r the needed
//     entries lie in MSF contiguous runs of `span` rows.  A wave reads such a run with ONE coalesced load (lane = row), keeps the
//     accepted rows (LDS map row -> index) in an LDS tile [i][32 columns], and the tile goes out as 512-byte runs of GrT.  Round 2
//     read the same sectors with one 16-byte request per lane and sector (1.41 x the algorithmic traffic at 0.30 of the HBM roof).
//   * X: the columns G[:, I_i] are contiguous, the MFMA operand fragments of X = G[:, I] W come straight from global memory
//     (16 lanes = 16 consecutive rows); wave w forms the 16 columns 16 w .. 16 w + 15 of the tile (operand roles swapped as in
//     k_zgemm so that the stores coalesce).
// __launch_bounds__(256, 2): with 512 registers on offer the compiler keeps MFMA accumulators in AGPRs and copies them to VGPRs
// and back on every trip of a loop with a run-time trip count (round 2: 48 v_accvgpr_read + 48 writes per 6 MFMAs in this kernel).
typedef double u_v4d __attribute__((ext_vector_type(4)));
#define GATHER_MAXSPAN 2048
__global__ __launch_bounds__(256, 2) void k_update_gather(DevModel dm, const DevUpdateState* __restrict__ us,
                                                           const cplx* __restrict__ G, const cplx* __restrict__ Wg,
                                                           cplx* __restrict__ X, cplx* __restrict__ GrT, size_t cs) {
    CHAIN(us); CHAIN(G); CHAIN(Wg); CHAIN(X); CHAIN(GrT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ONE round trip for the block's bookkeeping: the count and all DQMC_MAX_WDIM site slots are requested together (slots >= j hold
    // sites of earlier blocks and are never used); round 3 before: count -> first / last site -> the list, three dependent trips
    const int j = us->block_j;
    const int mysite = us->block_sites[tid & (DQMC_MAX_WDIM - 1)];
    if (j <= 0) return;
    const int MSF = dm.MSF, N = dm.N, ng = dm.ng, WD = MSF * dm.D;
    const int nI = MSF * j, nI8 = (nI + 7) & ~7;
    __shared__ int ssite[DQMC_MAX_WDIM];
    __shared__ int sI[DQMC_MAX_WDIM];
    __shared__ short smap[GATHER_MAXSPAN];                 // site - first site of the block  ->  index among the accepted, or -1
    __shared__ cplx tile[DQMC_MAX_WDIM][33];
    if (tid < DQMC_MAX_WDIM) ssite[tid] = mysite;
    __syncthreads();
    const int s_first = ssite[0], span = ssite[j - 1] - s_first + 1;
    for (int t = tid; t < nI; t += 256) sI[t] = ssite[t / MSF] + (t % MSF) * N;
    for (int t = tid; t < span; t += 256) smap[t] = -1;
    __syncthreads();
    for (int t = tid; t < j; t += 256) smap[ssite[t] - s_first] = (short)t;
    __syncthreads();
    const int r0 = blockIdx.x * 32;
    // ---- GrT, phase 1: coalesced runs of rows -> LDS tile ----
    {
        const int nchunk = (span + 63) >> 6;
        const int nitems = 32 * MSF * nchunk;              // (column c, band b, chunk q)
        constexpr int U = 8;                               // loads in flight per lane
        for (int base = wave * U; base < nitems; base += 4 * U) {
            cplx h[U];
            int li[U], cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = min(base + u, nitems - 1);
                const int c = item & 31, bq = item >> 5, b = bq % MSF, q = bq / MSF;
                const int t = q * 64 + lane;
                const int tcl = min(t, span - 1);
                const int row = b * N + s_first + tcl, col = min(r0 + c, ng - 1);
                h[u] = G[(size_t)col * ng + row];
                const int l = smap[tcl];
                const bool ok = base + u < nitems && t < span && l >= 0;
                li[u] = ok ? l * MSF + b : -1;
                cc[u] = c;
                if (row == r0 + c) h[u].x -= 1.0;          // - E_I: the unit entry sits where row I_i meets column I_i
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (li[u] >= 0) tile[li[u]][cc[u]] = h[u];
        }
    }
    __syncthreads();
    // ---- GrT, phase 2: 512-byte runs, zero padding up to nI8 ----
    {
        const int c = tid & 31, r = r0 + c;
        for (int i = tid >> 5; i < nI8; i += 8)
            if (r < ng) GrT[(size_t)i * ng + r] = (i < nI) ? tile[i][c] : make_double2(0.0, 0.0);
    }
    // ---- X tile: rows r0 .. r0 + 31, columns 16 wave .. 16 wave + 15 ----
    const int c0 = wave * 16;
    if (c0 >= nI8) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    // 3M complex product as in k_zgemm (kernels_gemm.hip): acc_re = P1 = w_r g_r, acc_p2 = P2 = w_i g_i, acc_im = P3 = (w_r + w_i)(g_r + g_i)
    u_v4d acc_re[2], acc_im[2], acc_p2[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) { acc_re[a] = (u_v4d)(0.0); acc_im[a] = (u_v4d)(0.0); acc_p2[a] = (u_v4d)(0.0); }
    // Operands of one k-step.  Only W carries a mask (k >= nI or a padding column: the product must be zero there); G is read at
    // clamped, always valid addresses and needs none -- a zero W entry annihilates whatever finite value sits there, rows past n_g
    // are never stored.  The mask is an AND on the bit pattern: a select on a loaded value compiles to a branch around the load
    // and a wait behind it.  Two operand sets in flight (ping-pong, written out twice: see k_flush).
    const unsigned long long colmask = (c0 + l15 < nI) ? ~0ull : 0ull;
    const cplx* wcol = Wg + (size_t)min(c0 + l15, nI - 1) * WD;
    const int ra = min(r0 + l15, ng - 1), rb = min(r0 + 16 + l15, ng - 1);
    auto ldk = [&](int k0, cplx& w, cplx (&g)[2]) {
        const int gk = k0 + l4;
        const int gkc = min(gk, nI - 1);
        const unsigned long long m = (gk < nI) ? colmask : 0ull;
        const cplx wl = wcol[gkc];
        w = make_double2(__longlong_as_double(__double_as_longlong(wl.x) & m), __longlong_as_double(__double_as_longlong(wl.y) & m));
        const cplx* gc = G + (size_t)sI[gkc] * ng;
        g[0] = gc[ra]; g[1] = gc[rb];
    };
    auto mack = [&](const cplx& w, const cplx (&g)[2]) {
        const double ws = w.x + w.y;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            acc_re[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x, g[a].x, acc_re[a], 0, 0, 0);
            acc_p2[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.y, g[a].y, acc_p2[a], 0, 0, 0);
            acc_im[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(ws, g[a].x + g[a].y, acc_im[a], 0, 0, 0);
        }
    };
    {
        // two operand sets in flight (ping-pong, written out twice: see k_flush).  A ring of four sets was measured and changed nothing
        // (385 vs 377 ms per 10 steps of 128 chains): the launch moves 275 MB of physical HBM traffic in 47 us (5.8 TB/s, PMC) -- it is
        // bound by the bytes it fetches, 1.39 x the algorithmic ones because of the row windows, not by the latency of this loop.
        cplx wA, wB, gA[2], gB[2];
        ldk(0, wA, gA);
        ldk(4, wB, gB);                                   // k >= nI: masked to zero
        __builtin_amdgcn_sched_barrier(0);
        for (int k0 = 0; k0 < nI; k0 += 8) {
            mack(wA, gA);
            __builtin_amdgcn_sched_barrier(0);
            ldk(k0 + 8, wA, gA);
            __builtin_amdgcn_sched_barrier(0);
            mack(wB, gB);
            __builtin_amdgcn_sched_barrier(0);
            ldk(k0 + 12, wB, gB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int c = c0 + l4 + 4 * rr, r = r0 + a * 16 + l15;      // D[m = l4 + 4 rr][n = l15]
            const double p1 = acc_re[a][rr], p2 = acc_p2[a][rr];
            // columns nI .. nI8 - 1: W was masked to zero there, so the product IS the zero padding
            if (c < nI8 && r < ng) X[(size_t)c * ng + r] = make_double2(p1 - p2, (acc_im[a][rr] - p1) - p2);
        }
}

// The proposal window of the NEXT delayed-update block, with this block's update already in it:
//   Gw[(b P + t') ldw + a P + t] = G[row, col] + sum_k X[row, k] GrT[col, k],   row = a N + s0 + t, col = b N + s0 + t',
// s0 = the site the next decision launch starts at, P = pbudget sites, ldw = MSF P -- the (MSF P)^2 entries the next k_update_decide
// launch can touch (see there).  With this copy the decisions of block b + 1 do not have to wait for the flush of block b over the
// whole of G: the flush runs on a second stream meanwhile.  Also publishes K = MSF j for that flush (flush_k: the next decision
// launch overwrites block_j while the flush may still be reading its K).  One thread per entry.
__global__ __launch_bounds__(256) void k_update_window(DevModel dm, DevUpdateState* us, const cplx* __restrict__ G,
                                                        const cplx* __restrict__ X, const cplx* __restrict__ GrT,
                                                        cplx* __restrict__ Gw, int P, size_t cs) {
    CHAIN(us); CHAIN(G); CHAIN(X); CHAIN(GrT); CHAIN(Gw);
    const int MSF = dm.MSF, N = dm.N, ng = dm.ng;
    const int j = us->block_j, s0 = us->site_cursor;
    if (blockIdx.x == 0 && threadIdx.x == 0) us->flush_k = MSF * j;
    if (us->slice_done || s0 >= N) return;
    const int K8 = (MSF * j + 7) & ~7;
    const int ldw = MSF * P;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ldw * ldw) return;
    const int iw = e % ldw, jw = e / ldw;
    const int a = iw / P, t = iw - a * P, b = jw / P, t2 = jw - b * P;
    if (s0 + t >= N || s0 + t2 >= N) return;               // past the last site: never read
    const int row = a * N + s0 + t, col = b * N + s0 + t2;
    cplx acc = G[(size_t)col * ng + row];
    const cplx* xr = X + row;
    const cplx* gc = GrT + col;
    for (int kk = 0; kk < K8; ++kk) acc = u_cfma(xr[(size_t)kk * ng], gc[(size_t)kk * ng], acc);
    Gw[(size_t)jw * ldw + iw] = acc;
}
void launch_update_window(const Launch& lc, const DevModel& hm, DevUpdateState* us, const cplx* G, const cplx* X, const cplx* GrT,
                          cplx* Gw, int P) {
    const int ldw = hm.MSF * P;
    hipLaunchKernelGGL(k_update_window, dim3((ldw * ldw + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, hm, us, G, X, GrT, Gw, P, lc.cs);
}

void launch_update_gather(const Launch& lc, const DevModel& hm, const DevUpdateState* us, const cplx* G,
                          const cplx* W, cplx* X, cplx* GrT) {
    hipLaunchKernelGGL(k_update_gather, dim3((hm.ng + 31) / 32, 1, lc.nb), dim3(256), 0, lc.st, hm, us, G, W, X, GrT, lc.cs);
}
