// Fermionic observables of one time slice, accumulated on the device from the shifted Green's function
// gs = e^{-dtau K/2} G e^{+dtau K/2} (shiftGreenSymmetric, reference src/detsdwopdim.cpp:4507-4612; the shift itself is
// two half-step plaquette passes per side in k_bmult_chain's shift mode, or two GEMMs with the dense half propagators).
//
// Replaces the G-dependent part of DetSDW::measure (src/detsdwopdim.cpp:545-899): greenK0 (sum of all entries),
// greenLocal (trace), occDiffSq, the pairing correlators pairPlus / pairMinus, and the momentum-space occupation
// kOccX / kOccY.  The reference evaluates kOcc with an O(N^3) loop of complex exponentials per slice; here a slice
// only bins  S_band(dx, dy) = sum_{i - j = (dx, dy)} [g_band,up(i, j) + g_band,down(i, j)]  over the (2L-1)^2 plain
// (not periodic: antiperiodic boundaries shift k by half a step) site differences, and the Fourier sum over the bins
// is done once per sweep on the host (detsdw.cpp).  Every accumulator has exactly one writer and a fixed summation
// order: results are reproducible bit for bit.
#include "dqmc_internal.h"

size_t measure_accum_doubles(int N, int L) {
    const size_t nbins = (size_t)(2 * L - 1) * (2 * L - 1);
    return 4 + 2 * (size_t)N + 4 * nbins;
}

__device__ __forceinline__ cplx m_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx m_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx m_mul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx m_scale(double s, cplx a) { return make_double2(s * a.x, s * a.y); }

// BandSpin: XUP = 0, YDOWN = 1, XDOWN = 2, YUP = 3 (detsdwopdim.h:223-232); gl1 of measure() (:594-612): for OPDIM < 3
// only the (XUP, YDOWN) sector is stored, the (XDOWN, YUP) sector is its complex conjugate, the rest vanishes
template<int OPDIM>
struct GreenAccess {
    const cplx* gs; int ng, N;
    __device__ __forceinline__ cplx operator()(int s1, int bs1, int s2, int bs2) const {
        if (OPDIM == 3) return gs[(size_t)(s2 + N * bs2) * ng + s1 + N * bs1];
        if (bs1 < 2 && bs2 < 2) return gs[(size_t)(s2 + N * bs2) * ng + s1 + N * bs1];
        if (bs1 >= 2 && bs2 >= 2) { cplx v = gs[(size_t)(s2 + N * (bs2 - 2)) * ng + s1 + N * (bs1 - 2)]; return make_double2(v.x, -v.y); }
        return make_double2(0.0, 0.0);
    }
};
// getBandSpin (detsdwopdim.h:268-273): band 0 = X, 1 = Y; spin 0 = up, 1 = down
__device__ __forceinline__ int m_bs(int band, int spin) { return band == 0 ? (spin == 0 ? 0 : 2) : (spin == 0 ? 3 : 1); }

__device__ __forceinline__ double block_sum(double v, double* red) {      // fixed-order tree over 256 threads
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

template<int OPDIM>
__global__ __launch_bounds__(256) void k_measure_accum(DevModel dm, const cplx* __restrict__ gs, double* __restrict__ acc, size_t cs) {
    CHAIN(gs); CHAIN(acc);
    __shared__ double red[256];
    const int N = dm.N, ng = dm.ng, L = dm.L, tid = threadIdx.x;
    const GreenAccess<OPDIM> g1{gs, ng, N};
    auto gl = [&](int s1, int b1, int sp1, int s2, int b2, int sp2) { return g1(s1, m_bs(b1, sp1), s2, m_bs(b2, sp2)); };
    const int role = blockIdx.x;
    constexpr int X = 0, Y = 1, UP = 0, DN = 1;
    if (role == 0) {
        // greenK0 += [2] Re sum(gs), greenLocal += [2] Re tr(gs) / (4 N)   (:565-588)
        double s = 0.0, t = 0.0;
        const size_t total = (size_t)ng * ng;
        {   // one workgroup sums n_g^2 elements: eight independent partial sums keep eight loads per thread in flight
            double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            size_t idx = tid;
            for (; idx + 7 * 256 < total; idx += 8 * 256) {
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] += gs[idx + (size_t)u * 256].x;
            }
            for (; idx < total; idx += 256) p[0] += gs[idx].x;
            s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        }
        for (int i = tid; i < ng; i += 256) t += gs[(size_t)i * ng + i].x;
        s = block_sum(s, red);
        t = block_sum(t, red);
        if (tid == 0) {
            const double f = (OPDIM == 3) ? 1.0 : 2.0;
            acc[0] += f * s;
            acc[1] += f * t / (4.0 * N);
            acc[3] += 1.0;
        }
    } else if (role == 1) {
        // occDiffSq (:866-897)
        double c = 0.0;
        for (int i = tid; i < N; i += 256) {
            auto g = [&](int b1, int s1, int b2, int s2) { return gl(i, b1, s1, i, b2, s2); };
            cplx v = m_scale(-2.0, m_mul(g(X, DN, X, UP), g(X, UP, X, DN)));
            v = m_add(v, g(X, UP, X, UP));
            v = m_add(v, m_scale(2.0, m_mul(g(X, DN, Y, DN), g(Y, DN, X, DN))));
            v = m_add(v, m_scale(2.0, m_mul(g(X, UP, Y, DN), g(Y, DN, X, UP))));
            v = m_add(v, g(Y, DN, Y, DN));
            v = m_sub(v, m_scale(2.0, m_mul(g(X, UP, X, UP), g(Y, DN, Y, DN))));
            v = m_add(v, m_scale(2.0, m_mul(g(X, DN, Y, UP), g(Y, UP, X, DN))));
            v = m_add(v, m_scale(2.0, m_mul(g(X, UP, Y, UP), g(Y, UP, X, UP))));
            v = m_sub(v, m_scale(2.0, m_mul(g(Y, DN, Y, UP), g(Y, UP, Y, DN))));
            cplx f = make_double2(1.0, 0.0);
            f = m_add(f, m_scale(2.0, g(X, UP, X, UP)));
            f = m_sub(f, m_scale(2.0, g(Y, DN, Y, DN)));
            f = m_sub(f, m_scale(2.0, g(Y, UP, Y, UP)));
            v = m_add(v, m_mul(g(X, DN, X, DN), f));
            v = m_add(v, g(Y, UP, Y, UP));
            v = m_sub(v, m_scale(2.0, m_mul(g(X, UP, X, UP), g(Y, UP, Y, UP))));
            v = m_add(v, m_scale(2.0, m_mul(g(Y, DN, Y, DN), g(Y, UP, Y, UP))));
            c += v.x;
        }
        c = block_sum(c, red);
        if (tid == 0) acc[2] += c / (double)N;
    } else if (role == 2) {
        // pairPlus[i], pairMinus[i] (:661-722): site pairs (i, 0) and (0, i)
        for (int i = tid; i < N; i += 256) {
            cplx plus = make_double2(0.0, 0.0), minus = make_double2(0.0, 0.0);
            for (int pr = 0; pr < 2; ++pr) {
                const int A = pr == 0 ? i : 0, B = pr == 0 ? 0 : i;
                auto P = [&](int b1, int b2) {
                    return m_sub(m_mul(gl(A, b1, DN, B, b2, UP), gl(A, b1, UP, B, b2, DN)),
                                 m_mul(gl(A, b1, DN, B, b2, DN), gl(A, b1, UP, B, b2, UP)));
                };
                const cplx pxx = P(X, X), pxy = P(X, Y), pyx = P(Y, X), pyy = P(Y, Y);
                plus = m_add(plus, m_scale(-4.0, m_add(m_add(m_add(pxx, pxy), pyx), pyy)));
                minus = m_add(minus, m_scale(-4.0, m_add(m_sub(m_sub(pxx, pxy), pyx), pyy)));
            }
            acc[4 + i] += plus.x;
            acc[4 + N + i] += minus.x;
        }
    } else {
        // S_band(dx, dy) bins for the momentum-space occupation (:616-659): 8 lanes per bin, each takes every 8th site,
        // partial sums combined in a fixed order by DPP row shifts (lane 7 of the group ends up with the total)
        const int W = 2 * L - 1, nbins = W * W;
        const int idx = (role - 3) * 256 + tid;
        const int binlin = idx >> 3, part = idx & 7;
        const bool valid = binlin < 2 * nbins;
        const int band = valid ? binlin / nbins : 0, bin = valid ? binlin - band * nbins : 0;
        const int dx = bin % W - (L - 1), dy = bin / W - (L - 1);
        cplx s = make_double2(0.0, 0.0);
        if (valid)
            for (int i = part; i < N; i += 8) {
                const int ix = i % L, iy = i / L;
                const int jx = ix - dx, jy = iy - dy;
                if (jx < 0 || jx >= L || jy < 0 || jy >= L) continue;
                const int j = jy * L + jx;
                s = m_add(s, m_add(gl(i, band, UP, j, band, UP), gl(i, band, DN, j, band, DN)));
            }
        auto shr_add = [](double v, int ctrl_sel) {
            int lo = __double2loint(v), hi = __double2hiint(v), lo2, hi2;
            if (ctrl_sel == 1) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); }
            else if (ctrl_sel == 2) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); }
            else { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); }
            return v + __hiloint2double(hi2, lo2);
        };
        s.x = shr_add(s.x, 1); s.y = shr_add(s.y, 1);      // row_shr:1, 2, 4: lane l holds the sum of lanes l-7 .. l
        s.x = shr_add(s.x, 2); s.y = shr_add(s.y, 2);
        s.x = shr_add(s.x, 4); s.y = shr_add(s.y, 4);
        if (valid && part == 7) {
            double* S = acc + 4 + 2 * N + (size_t)band * 2 * nbins;
            S[2 * bin] += s.x;
            S[2 * bin + 1] += s.y;
        }
    }
}

void launch_measure_accum(const Launch& lc, const DevModel& hm, const cplx* gs, double* acc) {
    const int nbins = (2 * hm.L - 1) * (2 * hm.L - 1);
    const dim3 grid(3 + (2 * nbins * 8 + 255) / 256, 1, lc.nb);
    if (hm.opdim == 1) hipLaunchKernelGGL((k_measure_accum<1>), grid, dim3(256), 0, lc.st, hm, gs, acc, lc.cs);
    else if (hm.opdim == 2) hipLaunchKernelGGL((k_measure_accum<2>), grid, dim3(256), 0, lc.st, hm, gs, acc, lc.cs);
    else hipLaunchKernelGGL((k_measure_accum<3>), grid, dim3(256), 0, lc.st, hm, gs, acc, lc.cs);
}
