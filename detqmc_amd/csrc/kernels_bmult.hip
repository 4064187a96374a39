// Checkerboard B-matrix multiplication chains for the SDW model on gfx950.
//
// Replaces DetSDW::checkerboard{Left,Right}MultiplyBmat[Inv] and the per-slice
// leftMultiplyBk / leftMultiplyBkInv / rightMultiplyBk / rightMultiplyBkInv
// (reference src/detsdwopdim.cpp:1996-2420) including the plaquette passes
// cb_assaad_applyBondFactors{Left,Right}[_precalcedMatrices] (:1688-1756, :1788-1826, :1905-1943).
//
// B_k = e^{-dtau V(phi_k)} . diag(e^{dtau mu_band}) . e^{-dtau K}, e^{-dtau K} in the symmetric
// break-up e^{-dtau K1/2} e^{-dtau K0} e^{-dtau K1/2} per N x N band block.
//
// Design (HBM/L2-bound streaming op, no MFMA): a LEFT multiply acts on every column of A
// independently, a RIGHT multiply on every row.  One workgroup stages a few whole vectors
// (columns resp. rows) in LDS, applies the whole chain of slices k to them there -- the s slices of
// a UdV chain cost ONE read and ONE write of A instead of s -- and streams them back.  Column
// vectors are contiguous in the column-major matrix; row vectors are staged as tiles of 8 rows (one full 128-byte line of every column,
// fewer rows when 9 rows of n_g complex numbers do not fit the 144 KiB LDS budget).
#include "dqmc_internal.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) {
    c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
    c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
    return c;
}
__device__ __forceinline__ cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }

// e^{sign dtau V} at one site: entries as in get_delta_forsite's evMatrix (detsdwopdim.cpp:3188-3229) == the vectors
// cd/cmd/mbx/mbcx/ax/max of :2001-2030.  c0 / c1 / xs from cdw_site_terms (c0 = c1 = cosh term while cdwU == 0).
template<int MSF>
__device__ __forceinline__ void build_V(cplx (&V)[MSF][MSF], double sign, double c0, double c1, double xs,
                                        double p0, double p1, double p2) {
#pragma unroll
    for (int a = 0; a < MSF; ++a)
#pragma unroll
        for (int b = 0; b < MSF; ++b) V[a][b] = make_double2(0.0, 0.0);
    cplx bx = make_double2(sign * p0 * xs, -sign * p1 * xs);   // sign (phi0 - i phi1) x
    cplx bcx = make_double2(sign * p0 * xs, sign * p1 * xs);   // sign (phi0 + i phi1) x
    V[0][0] = make_double2(c0, 0.0);
    V[1][1] = make_double2(c1, 0.0);
    V[0][1] = bx;
    V[1][0] = bcx;
    if (MSF == 4) {
        double ax = sign * p2 * xs;
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

// CDW: cdwU != 0 (the discrete field's terms in e^{dtau V}); a template parameter so that the cdwU == 0 kernels are the round-2 code
template<int MSF, bool RIGHT, bool INV, bool CDW>
__global__ __launch_bounds__(512, MSF == 2 ? 4 : 2) void k_bmult_chain(DevModel dm, cplx* __restrict__ A, int lda, int nvec,
                                                      int kfirst, int kstep, int kcount, int shift, size_t cs) {
    extern __shared__ cplx sm[];
    dm = chain_model(dm, cs); CHAIN(A);
    const int N = dm.N, ng = dm.ng, P = dm.P;
    const int v0 = blockIdx.x * nvec;
    const int nv = min(nvec, ng - v0);
    if (nv <= 0) return;
    const int tid = threadIdx.x, nth = blockDim.x;
    // LDS address of element e of vector v
    // RIGHT: rows are staged as pieces of nvec elements per column; one padding element per column keeps lanes that
    // work on sites 2 apart off the same LDS banks
    const int rstride = nvec + 1;
    auto addr = [&](int v, int e) -> int { return RIGHT ? (e * rstride + v) : (v * ng + e); };

    // ---- stage in ----
    // (the same digit-wise stepping as for the work items further down: no division per element)
    const int rv0 = tid % nvec, re0 = tid / nvec, rvd = nth % nvec, red = nth / nvec;
    // SB elements per thread are requested before the first of them is written to LDS (left as `sm[idx] = src[idx]` the loop
    // compiles to load, s_waitcnt vmcnt(0), ds_write per element: eight dependent trips to HBM per tile); nontemporal: A is
    // streamed through once, the tables of the passes below stay in the L2
    constexpr int SB = 8;
    if (!RIGHT) {
        const int total = nv * ng;
        const cplx* src = A + (size_t)v0 * lda;
        for (int base = tid; base < total; base += SB * nth) {
            cplx r[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * nth, total - 1);
                size_t off = idx;                                        // lda == ng: the nv columns are one contiguous run
                if (lda != ng) { const int v = idx / ng; off = (size_t)v * lda + (idx - v * ng); }
                r[u] = nt_load(src + off);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) { const int idx = base + u * nth; if (idx < total) sm[idx] = r[u]; }
        }
    } else {
        int v = rv0, e = re0;
        const int total = nvec * ng;
        for (int base = tid; base < total; base += SB * nth) {
            cplx r[SB];
            int la[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                la[u] = (base + u * nth < total && v < nv) ? e * rstride + v : -1;
                r[u] = nt_load(A + (size_t)min(e, ng - 1) * lda + (v0 + min(v, nv - 1)));
                v += rvd; e += red;
                if (v >= nvec) { v -= nvec; e += 1; }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) if (la[u] >= 0) sm[la[u]] = r[u];
        }
    }
    __syncthreads();

    constexpr bool PASSES_FIRST = (RIGHT == INV);   // left B, right B^-1: hopping part acts first
    const int signIdx = INV ? 1 : 0;
    const double vsign = INV ? +1.0 : -1.0;

    // Work-item decomposition without a run-time division per item and pass: an item index idx = tid, tid + nth, ... is a
    // mixed-radix number; its digits for idx = tid and for the increment nth are computed ONCE per launch, the loops below add
    // digit-wise with carries.  (Measured: 79.3 -> 78.2 ms per 128-chain sweep -- the SQ counters show 1 170 VALU instructions
    // per wave and launch, but the divisions were not the bulk of them.)
    //   plaquette items   LEFT: (p, t = v MSF + b) radices (P, -);   RIGHT: (v, p, b) radices (nv, P, -)
    //   site items        (i, v) radices (N, -)
    int pl0[3], pld[3];
    if (!RIGHT) { pl0[0] = tid % P; pl0[1] = tid / P; pl0[2] = 0; pld[0] = nth % P; pld[1] = nth / P; pld[2] = 0; }
    else {
        pl0[0] = tid % nv; const int t0 = tid / nv; pl0[1] = t0 % P; pl0[2] = t0 / P;
        pld[0] = nth % nv; const int t1 = nth / nv; pld[1] = t1 % P; pld[2] = t1 / P;
    }
    const int si0 = tid % N, sv0 = tid / N, sid = nth % N, svd = nth / N;

    for (int kc = 0; kc < kcount; ++kc) {
        const int k = kfirst + kc * kstep;
        for (int stage = 0; stage < 2; ++stage) {
            const bool do_passes = (stage == 0) == PASSES_FIRST;
            if (shift && !do_passes) continue;           // shiftGreenSymmetric: hopping half steps only, no e^{-+dtau V}
            if (do_passes) {
                if (dm.dense) continue;      // CB_NONE: dense e^{+-dtau K} applied by a GEMM outside this kernel
                // e^{+-dtau K1/2} e^{+-dtau K0} e^{+-dtau K1/2}: sub 1 (half), sub 0 (full), sub 1 (half)
                // shift (shiftGreenSymmetric, detsdwopdim.cpp:4527-4541): e^{+-dtau K1/2} then e^{+-dtau K0/2}, half-step tables
                const double* abcd_tab = shift ? dm.pabcd_h : dm.pabcd;
                const cplx* mat_tab = shift ? dm.pmats_h : dm.pmats;
                const int npass = shift ? 2 : 3;
                for (int pass = 0; pass < npass; ++pass) {
                    const int sub = shift ? (pass == 0 ? 1 : 0) : ((pass == 1) ? 0 : 1);
                    // The site-local e^{-+dtau V} mix is FUSED into the plaquette pass next to it (the first pass when V acts first,
                    // the last one otherwise): a work item then holds all MSF band entries of its four sites, so the mix happens in
                    // registers between the LDS read and the LDS write of that pass -- one LDS round trip and one barrier per slice
                    // less (the kernel's compute phase is bound by the LDS pipe: 4 reads + 4 writes of 16 B per element and slice
                    // before, SQ_INSTS_LDS / SQ_LDS_BANK_CONFLICT in profiles/r02_pmc_traffic_b128_d32.json).  Per element the
                    // arithmetic and its order are those of the separate stages.
                    if (!shift && pass == (PASSES_FIRST ? npass - 1 : 0)) {
                        const int fitems = nv * P;
                        const double* ph = dm.phi + (size_t)k * dm.opdim * N;
                        for (int idx = tid; idx < fitems; idx += nth) {
                            int p, v;
                            if (RIGHT) { v = idx % nv; p = idx / nv; } else { p = idx % P; v = idx / P; }
                            int site[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) site[q] = dm.psites[(sub * 4 + q) * P + p];
                            cplx x[MSF][4];
#pragma unroll
                            for (int b = 0; b < MSF; ++b)
#pragma unroll
                                for (int q = 0; q < 4; ++q) x[b][q] = sm[addr(v, b * N + site[q])];
                            auto vmix = [&]() {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int i = site[q];
                                    const double c = dm.coshT[(size_t)k * N + i];
                                    double xs = dm.sinhT[(size_t)k * N + i], c0, c1;
                                    if constexpr (CDW) cdw_site_terms(dm, (size_t)k * N + i, vsign, c, c0, c1, xs); else { c0 = c; c1 = c; }
                                    const double p0 = ph[i];
                                    const double p1 = dm.opdim > 1 ? ph[N + i] : 0.0;
                                    const double p2 = dm.opdim > 2 ? ph[2 * N + i] : 0.0;
                                    cplx V[MSF][MSF];
                                    build_V<MSF>(V, vsign, c0, c1, xs, p0, p1, p2);
                                    cplx in[MSF], out[MSF];
#pragma unroll
                                    for (int b = 0; b < MSF; ++b) in[b] = x[b][q];
                                    if (PASSES_FIRST) {
#pragma unroll
                                        for (int b = 0; b < MSF; ++b) in[b] = cscale(in[b], INV ? dm.ovinv[b & 1] : dm.ov[b & 1]);
                                    }
#pragma unroll
                                    for (int o = 0; o < MSF; ++o) {
                                        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                                        for (int b = 0; b < MSF; ++b) {
                                            const cplx vv = RIGHT ? V[b][o] : V[o][b];
                                            acc = cfma(vv, in[b], acc);
                                        }
                                        if (!PASSES_FIRST) acc = cscale(acc, INV ? dm.ovinv[o & 1] : dm.ov[o & 1]);
                                        out[o] = acc;
                                    }
#pragma unroll
                                    for (int b = 0; b < MSF; ++b) x[b][q] = out[b];
                                }
                            };
                            if (!PASSES_FIRST) vmix();
#pragma unroll
                            for (int b = 0; b < MSF; ++b) {
                                const int tbl = ((b & 1) * 2 + signIdx) * 2 + sub;
                                cplx y[4];
                                if (dm.pm_real) {
                                    double co[4];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) co[q] = abcd_tab[(tbl * 4 + q) * P + p];
#pragma unroll
                                    for (int a = 0; a < 4; ++a) {
                                        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                                        for (int q = 0; q < 4; ++q) {
                                            const double m = co[a ^ q];
                                            acc.x = fma(m, x[b][q].x, acc.x);
                                            acc.y = fma(m, x[b][q].y, acc.y);
                                        }
                                        y[a] = acc;
                                    }
                                } else {
                                    const cplx* mat = mat_tab + (size_t)tbl * 16 * P + p;
#pragma unroll
                                    for (int a = 0; a < 4; ++a) {
                                        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                                        for (int q = 0; q < 4; ++q) {
                                            const cplx mm = RIGHT ? mat[(size_t)(q * 4 + a) * P] : mat[(size_t)(a * 4 + q) * P];
                                            acc = cfma(mm, x[b][q], acc);
                                        }
                                        y[a] = acc;
                                    }
                                }
#pragma unroll
                                for (int q = 0; q < 4; ++q) x[b][q] = y[q];
                            }
                            if (PASSES_FIRST) vmix();
#pragma unroll
                            for (int b = 0; b < MSF; ++b)
#pragma unroll
                                for (int q = 0; q < 4; ++q) sm[addr(v, b * N + site[q])] = x[b][q];
                        }
                        __syncthreads();
                        continue;
                    }
                    const int items = nv * MSF * P;
                    int d0 = pl0[0], d1 = pl0[1], d2 = pl0[2];
                    for (int idx = tid; idx < items; idx += nth) {
                        int p, b, v;
                        if (RIGHT) { v = d0; p = d1; b = d2; }                                   // vectors fastest: contiguous in LDS
                        else { p = d0; b = d1 % MSF; v = d1 / MSF; }                             // MSF: compile time
                        // next item of this thread
                        d0 += pld[0]; d1 += pld[1]; d2 += pld[2];
                        if (RIGHT) { if (d0 >= nv) { d0 -= nv; d1 += 1; } if (d1 >= P) { d1 -= P; d2 += 1; } }
                        else { if (d0 >= P) { d0 -= P; d1 += 1; } }
                        int band = b & 1;
                        const int tbl = (band * 2 + signIdx) * 2 + sub;
                        int e[4];
                        cplx x[4], y[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { e[q] = addr(v, b * N + dm.psites[(sub * 4 + q) * P + p]); x[q] = sm[e[q]]; }
                        if (dm.pm_real) {
                            // rows (a b c d)(b a d c)(c d a b)(d c b a): symmetric, so left and right agree
                            double co[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) co[q] = abcd_tab[(tbl * 4 + q) * P + p];
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const double m = co[a ^ q];       // entry (a, q) of the pattern above
                                    acc.x = fma(m, x[q].x, acc.x);
                                    acc.y = fma(m, x[q].y, acc.y);
                                }
                                y[a] = acc;
                            }
                        } else {
                            const cplx* mat = mat_tab + (size_t)tbl * 16 * P + p;
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    cplx mm = RIGHT ? mat[(size_t)(q * 4 + a) * P] : mat[(size_t)(a * 4 + q) * P];
                                    acc = cfma(mm, x[q], acc);
                                }
                                y[a] = acc;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) sm[e[q]] = y[q];
                    }
                    __syncthreads();
                }
            } else {
                // potential part: per-site MSF x MSF mix (+ chemical potential factor of the band index
                // that faces the hopping part).  Checkerboard: fused into the neighbouring plaquette pass above; this stage
                // only runs next to a dense propagator (CB_NONE).
                if (!dm.dense) continue;
                const int items = nv * N;
                const double* ph = dm.phi + (size_t)k * dm.opdim * N;
                int i = si0, v = sv0, inext, vnext;
                for (int idx = tid; idx < items; idx += nth, i = inext, v = vnext) {
                    inext = i + sid; vnext = v + svd;
                    if (inext >= N) { inext -= N; vnext += 1; }
                    double c = dm.coshT[(size_t)k * N + i], xs = dm.sinhT[(size_t)k * N + i], c0, c1;
                    if constexpr (CDW) cdw_site_terms(dm, (size_t)k * N + i, vsign, c, c0, c1, xs); else { c0 = c; c1 = c; }
                    double p0 = ph[i];
                    double p1 = dm.opdim > 1 ? ph[N + i] : 0.0;
                    double p2 = dm.opdim > 2 ? ph[2 * N + i] : 0.0;
                    cplx V[MSF][MSF];
                    build_V<MSF>(V, vsign, c0, c1, xs, p0, p1, p2);
                    cplx in[MSF], out[MSF];
#pragma unroll
                    for (int q = 0; q < MSF; ++q) in[q] = sm[addr(v, q * N + i)];
                    // input-side factor: left B and right B^-1 have the hopping part on the input side
                    if (PASSES_FIRST) {
#pragma unroll
                        for (int q = 0; q < MSF; ++q) in[q] = cscale(in[q], INV ? dm.ovinv[q & 1] : dm.ov[q & 1]);
                    }
#pragma unroll
                    for (int o = 0; o < MSF; ++o) {
                        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
                        for (int q = 0; q < MSF; ++q) {
                            cplx vv = RIGHT ? V[q][o] : V[o][q];
                            acc = cfma(vv, in[q], acc);
                        }
                        if (!PASSES_FIRST) acc = cscale(acc, INV ? dm.ovinv[o & 1] : dm.ov[o & 1]);
                        out[o] = acc;
                    }
#pragma unroll
                    for (int q = 0; q < MSF; ++q) sm[addr(v, q * N + i)] = out[q];
                }
                __syncthreads();
            }
        }
    }

    // ---- stage out ----
    if (!RIGHT) {
        cplx* dst = A + (size_t)v0 * lda;
        for (int idx = tid; idx < nv * ng; idx += nth) {
            size_t off = idx;
            if (lda != ng) { const int v = idx / ng; off = (size_t)v * lda + (idx - v * ng); }
            nt_store(dst + off, sm[idx]);
        }
    } else {
        int v = rv0, e = re0;
        for (int idx = tid; idx < nvec * ng; idx += nth) {
            if (v < nv) nt_store(A + (size_t)e * lda + (v0 + v), sm[e * rstride + v]);
            v += rvd; e += red;
            if (v >= nvec) { v -= nvec; e += 1; }
        }
    }
}

void launch_bmult(const Launch& lc, const DevModel* /*dm*/, const DevModel& hm, int side, int inverse,
                  int kfirst, int kstep, int kcount, cplx* A, int lda, int shift) {
    const int ng = hm.ng;
    // LEFT: <= 64 KiB of LDS; RIGHT: up to 144 KiB (one workgroup per CU then, two below 80 KiB) so that a row tile can be
    // 8 rows = one full 128-byte line of every column
    const size_t lds_cap = side == DQMC_LEFT ? 65536 : 144 * 1024;
    const int max_fit = (int)(lds_cap / ((size_t)ng * sizeof(cplx))) - (side == DQMC_LEFT ? 0 : 1);
    static const int env_l = dev_knob("DQMC_BMULT_NVEC_L") ? atoi(dev_knob("DQMC_BMULT_NVEC_L")) : 0;   // developer knobs
    static const int env_r = dev_knob("DQMC_BMULT_NVEC_R") ? atoi(dev_knob("DQMC_BMULT_NVEC_R")) : 0;
    int nvec;
    if (side == DQMC_LEFT) {
        // aim at >= 256 workgroups, but give every thread a work item in the fused plaquette + V pass (nv P items, P = N / 4):
        // L = 16: 4 columns per workgroup (single-slice launch of 128 chains 248 -> 219 us with the fused pass; 2 columns: 268)
        const int per_pass = (256 + hm.P - 1) / hm.P;
        nvec = env_l ? env_l : std::max(ng / 256, per_pass);
        if (nvec > ng) nvec = ng;
    } else {
        // row tiles of 8 rows: every global transaction of the strided row access is a FULL 128-byte line.  With 4-row
        // tiles (64-byte pieces) the other half of each line belongs to the neighbouring workgroup, which runs on another
        // XCD: rocprofv3 FETCH_SIZE showed 2.0x the matrix per launch (profiles/r02_pmc_traffic_*.json)
        nvec = env_r ? env_r : 8;
    }
    if (nvec > max_fit) nvec = max_fit;
    if (nvec < 1) nvec = 1;
    const int grid = (ng + nvec - 1) / nvec;
    const size_t lds = (size_t)(side == DQMC_LEFT ? nvec : nvec + 1) * ng * sizeof(cplx);
    if (lds > 48 * 1024) {      // raise the dynamic-LDS limit of the instantiation once per device
        static std::mutex mu;
        static size_t raised_tab[64][16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int slot = (hm.cdw_on ? 8 : 0) + (hm.MSF == 4 ? 4 : 0) + (side == DQMC_LEFT ? 0 : 2) + (inverse ? 1 : 0);
        std::lock_guard<std::mutex> lk(mu);
        size_t& raised = raised_tab[dev & 63][slot];
        if (lds > raised) {
            const void* f = nullptr;
#define BM_F(MSFV, R, I) (hm.cdw_on ? (const void*)k_bmult_chain<MSFV, R, I, true> : (const void*)k_bmult_chain<MSFV, R, I, false>)
            if (hm.MSF == 2) f = side == DQMC_LEFT ? (inverse ? BM_F(2, false, true) : BM_F(2, false, false)) : (inverse ? BM_F(2, true, true) : BM_F(2, true, false));
            else             f = side == DQMC_LEFT ? (inverse ? BM_F(4, false, true) : BM_F(4, false, false)) : (inverse ? BM_F(4, true, true) : BM_F(4, true, false));
#undef BM_F
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) raised = lds;
            else (void)hipGetLastError();      // the launch below then reports the problem
        }
    }
    // 8-row tiles hold twice the work of a 4-row tile and leave room for two workgroups per CU: 512 threads keep the
    // number of resident waves (the only thing that hides the LDS / memory latency of this streaming kernel) the same
    static const int env_t = dev_knob("DQMC_BMULT_THREADS_R") ? atoi(dev_knob("DQMC_BMULT_THREADS_R")) : 0;
    const int nthreads = (side == DQMC_LEFT) ? 256 : (env_t ? env_t : (nvec >= 8 ? 512 : 256));
#define LAUNCH(MSFV, R, I)                                                                              \
    do { if (hm.cdw_on) hipLaunchKernelGGL((k_bmult_chain<MSFV, R, I, true>), dim3(grid, 1, lc.nb), dim3(nthreads), lds, lc.st, hm, A, lda, nvec, \
                                           kfirst, kstep, kcount, shift, lc.cs);                        \
         else hipLaunchKernelGGL((k_bmult_chain<MSFV, R, I, false>), dim3(grid, 1, lc.nb), dim3(nthreads), lds, lc.st, hm, A, lda, nvec, \
                                 kfirst, kstep, kcount, shift, lc.cs); } while (0)
    if (hm.MSF == 2) {
        if (side == DQMC_LEFT) { if (!inverse) LAUNCH(2, false, false); else LAUNCH(2, false, true); }
        else                   { if (!inverse) LAUNCH(2, true, false);  else LAUNCH(2, true, true); }
    } else {
        if (side == DQMC_LEFT) { if (!inverse) LAUNCH(4, false, false); else LAUNCH(4, false, true); }
        else                   { if (!inverse) LAUNCH(4, true, false);  else LAUNCH(4, true, true); }
    }
#undef LAUNCH
}

// ---------------------------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------------------------
// updateCoshSinhTermsPhi (detsdwopdim.cpp:1132-1136, 1175-1181)
__global__ void k_cosh_sinh(DevModel dm, size_t cs) {
    dm = chain_model(dm, cs);
    const int total = dm.m * dm.N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int k = 1 + idx / dm.N, i = idx % dm.N;
        double nn = 0.0;
        for (int d = 0; d < dm.opdim; ++d) {
            double p = dm.phi[((size_t)k * dm.opdim + d) * dm.N + i];
            nn += p * p;
        }
        double nrm = sqrt(nn);
        double a = dm.lambda * dm.dtau * nrm;
        dm.coshT[(size_t)k * dm.N + i] = cosh(a);
        dm.sinhT[(size_t)k * dm.N + i] = sinh(a) / nrm;
    }
}
void launch_cosh_sinh(const Launch& lc, const DevModel& hm) {
    int total = hm.m * hm.N;
    hipLaunchKernelGGL(k_cosh_sinh, dim3((total + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, hm, lc.cs);
}

// updateCoshSinhTermsCDWl (detsdwopdim.cpp:1138-1143, 1183-1190): the caches of the discrete field, from the host-computed table of
// cosh / sinh(sqrt(dtau) cdwU eta(l)) (cosh even, sinh odd in l)
__global__ void k_cdw_terms(DevModel dm, size_t cs) {
    dm = chain_model(dm, cs);
    const int total = dm.m * dm.N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const size_t e = (size_t)dm.N + idx;                     // slices 1..m
        const double l = dm.cdwl[e];
        const int a = (fabs(l) > 1.5) ? 1 : 0;
        dm.cdwC[e] = dm.cdw_cosh[a];
        dm.cdwS[e] = (l < 0.0) ? -dm.cdw_sinh[a] : dm.cdw_sinh[a];
    }
}
void launch_cdw_terms(const Launch& lc, const DevModel& hm) {
    if (!hm.cdw_on) return;
    int total = hm.m * hm.N;
    hipLaunchKernelGGL(k_cdw_terms, dim3((total + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, hm, lc.cs);
}

__global__ void k_set_identity(cplx* A, int n, size_t cs) {
    CHAIN(A);
    size_t total = (size_t)n * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % n), j = (int)(idx / n);
        A[idx] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
}
void launch_set_identity(const Launch& lc, cplx* A, int n) {
    const int blocks = (int)std::min<size_t>(1024, ((size_t)n * n + 255) / 256);
    hipLaunchKernelGGL(k_set_identity, dim3(blocks, 1, lc.nb), dim3(256), 0, lc.st, A, n, lc.cs);
}

// B = A^H through a 32x32 LDS tile (both sides coalesced)
__global__ void k_conj_transpose(const cplx* __restrict__ A, cplx* __restrict__ B, int n, size_t cs) {
    __shared__ cplx tile[32][33];
    CHAIN(A); CHAIN(B);
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        int i = bx + tx, j = by + r;
        if (i < n && j < n) tile[r][tx] = A[(size_t)j * n + i];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int i = by + tx, j = bx + r;     // B(i, j) = conj(A(j, i))
        if (i < n && j < n) {
            cplx t = tile[tx][r];
            B[(size_t)j * n + i] = make_double2(t.x, -t.y);
        }
    }
}
void launch_conj_transpose(const Launch& lc, const cplx* A, cplx* B, int n) {
    dim3 grid((n + 31) / 32, (n + 31) / 32, lc.nb);
    hipLaunchKernelGGL(k_conj_transpose, grid, dim3(256), 0, lc.st, A, B, n, lc.cs);
}

__global__ void k_add_diag(cplx* A, const double* d, int n, size_t cs) {
    CHAIN(A); CHAIN(d);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * n + i].x += d[i];
}
void launch_add_diag(const Launch& lc, cplx* A, const double* d, int n) {
    hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, A, d, n, lc.cs);
}

// the same buffer of every chain (bytes must be a multiple of 8: all per-chain buffers are doubles / cplx)
__global__ void k_copy64(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, size_t words, size_t cs) {
    CHAIN(src); CHAIN(dst);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_bytes(const Launch& lc, const void* src, void* dst, size_t bytes) {
    if (lc.nb == 1) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, lc.st); return; }
    const size_t words = bytes / 8;
    const int blocks = (int)std::min<size_t>(2048, (words + 255) / 256);
    hipLaunchKernelGGL(k_copy64, dim3(blocks, 1, lc.nb), dim3(256), 0, lc.st, (const unsigned long long*)src,
                       (unsigned long long*)dst, words, lc.cs);
}
void launch_copy(const Launch& lc, const cplx* A, cplx* B, size_t count) {
    launch_copy_bytes(lc, A, B, count * sizeof(cplx));
}

// sum over slices 1..m, sites and components of phi^2 (get_exchange_action_contribution,
// detsdwopdim.cpp:5205-5216); single workgroup, fixed summation order => reproducible
__global__ void k_phi_sq_sum(DevModel dm, double* out, size_t cs) {
    __shared__ double red[256];
    dm = chain_model(dm, cs); CHAIN(out);
    double acc = 0.0;
    size_t total = (size_t)dm.m * dm.opdim * dm.N;
    const double* p = dm.phi + (size_t)dm.opdim * dm.N;   // skip slice 0
    for (size_t idx = threadIdx.x; idx < total; idx += 256) acc += p[idx] * p[idx];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}
// phiAction (detsdwopdim.cpp:4242-4300): the bosonic action of the whole field, one workgroup per chain, fixed summation
// order (per-thread partial sums over (slice, site) pairs in stride, then a tree) => reproducible.  r is the chain's own
// exchange parameter (DevUpdateState::r).
__global__ void k_phi_action(DevModel dm, const DevUpdateState* __restrict__ us, double* out, size_t cs) {
    __shared__ double red[256];
    dm = chain_model(dm, cs); CHAIN(us); CHAIN(out);
    const int N = dm.N, L = dm.L, m = dm.m, OPD = dm.opdim;
    const double dtau = dm.dtau, r = us->r, u = dm.u, c = dm.c;
    double acc = 0.0;
    for (int idx = threadIdx.x; idx < m * N; idx += 256) {
        const int k = 1 + idx / N, site = idx % N;
        const int kprev = (k > 1) ? k - 1 : m;
        const int x = site % L, y = site / L;
        const int xn = y * L + (x + 1) % L, yn = ((y + 1) % L) * L + x;
        double phisq = 0.0, td2 = 0.0, xd2 = 0.0, yd2 = 0.0;
        for (int d = 0; d < OPD; ++d) {
            const double ph = dm.phi[((size_t)k * OPD + d) * N + site];
            const double td = (ph - dm.phi[((size_t)kprev * OPD + d) * N + site]) / dtau;
            const double xd = ph - dm.phi[((size_t)k * OPD + d) * N + xn];
            const double yd = ph - dm.phi[((size_t)k * OPD + d) * N + yn];
            td2 += td * td; xd2 += xd * xd; yd2 += yd * yd; phisq += ph * ph;
        }
        double a = 0.5 * dtau * r * phisq;
        if (!dm.phi2bosons) a += (dtau / (2.0 * c * c)) * td2 + 0.5 * dtau * xd2 + 0.5 * dtau * yd2 + 0.25 * dtau * u * phisq * phisq;
        acc += a;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}
void launch_phi_action(const Launch& lc, const DevModel& hm, const DevUpdateState* us, double* out) {
    hipLaunchKernelGGL(k_phi_action, dim3(1, 1, lc.nb), dim3(256), 0, lc.st, hm, us, out, lc.cs);
}
// addGlobalRandomDisplacement (detsdwopdim.cpp:3755-3763): every slice (incl. the unused slice 0) of component d shifted by
// shifts[d], per chain
__global__ void k_phi_shift(DevModel dm, const double* __restrict__ shifts, size_t cs) {
    dm = chain_model(dm, cs);
    const double* sh = shifts + (size_t)blockIdx.z * dm.opdim;
    const int total = (dm.m + 1) * dm.opdim * dm.N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        dm.phi[idx] += sh[(idx / dm.N) % dm.opdim];
}
void launch_phi_shift(const Launch& lc, const DevModel& hm, const double* shifts) {
    const int total = (hm.m + 1) * hm.opdim * hm.N;
    hipLaunchKernelGGL(k_phi_shift, dim3((total + 255) / 256, 1, lc.nb), dim3(256), 0, lc.st, hm, shifts, lc.cs);
}
void launch_phi_sq_sum(const Launch& lc, const DevModel& hm, double* out) {
    hipLaunchKernelGGL(k_phi_sq_sum, dim3(1, 1, lc.nb), dim3(256), 0, lc.st, hm, out, lc.cs);
}
// dst[b] = factor * (first double of chain b's buffer `src`): the per-chain scalars of a batch gathered into ONE contiguous device
// array (the send buffer of a collective)
__global__ void k_gather_scalars(const double* __restrict__ src, size_t cs, int nb, double factor, double* __restrict__ dst) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) dst[b] = factor * *(const double*)((const char*)src + (size_t)b * cs);
}
void launch_gather_scalars(const Launch& lc, const double* src, double factor, double* dst) {
    hipLaunchKernelGGL(k_gather_scalars, dim3((lc.nb + 63) / 64), dim3(64), 0, lc.st, src, lc.cs, lc.nb, factor, dst);
}
