// Hubbard replica on gfx950 (BASELINE config 1; reference src/dethubbard.{h,cpp}).
//
// The reference keeps two real N x N Green's functions (spin up / down: DetModelGC<2, double, false>).  Here both live
// in ONE n_g = 2N block-diagonal matrix G = blockdiag(G_up, G_dn) stored like every other matrix of this library
// (complex fp64, imaginary parts zero), so that the whole stabilisation machinery -- dense-propagator B-multiplies on
// the MFMA GEMM, UdV / UDT factorisations, greenFromUdV, wrap / advance -- is shared with the SDW model:
//   B_k = blockdiag( diag(e^{+alpha s_k}) P, diag(e^{-alpha s_k}) P ),   P = proptmat = e^{-dtau T}   (dethubbard.cpp:823-849)
// What is specific to the model is in this file:
//   k_hubbard_vscale   the site-diagonal factor diag(e^{+-alpha s}) of B_k^{+-1} applied to rows (left) / columns (right)
//   k_hubbard_slice    updateInSlice (:141-172): N single-spin-flip proposals at RANDOM sites (randInt, with replacement),
//                      weightRatioSingleFlip (:858-874), Metropolis, rank-1 Sherman-Morrison update of both spin blocks
//                      (updateGreenFunctionWithFlip, :877-906) -- one workgroup per replica walks the whole slice
//   k_hubbard_measure  measure(timeslice) (:521-545): sums over G_ii, nearest-neighbour G_ij and the spin-z correlation
#include "dqmc_internal.h"

__global__ void k_hubbard_vscale(DevModel dm, cplx* __restrict__ A, int lda, int right, double sgn, int k, size_t cs) {
    dm = chain_model(dm, cs); CHAIN(A);
    const int N = dm.N, ng = dm.ng;
    const double ep = dm.hub_exp_alpha[0], em = dm.hub_exp_alpha[1];       // e^{+alpha}, e^{-alpha}
    const double* s = dm.phi + (size_t)k * N;
    const size_t total = (size_t)ng * ng;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % ng), j = (int)(idx / ng);
        const int e = right ? j : i;
        const double spin = (e < N) ? +1.0 : -1.0;                           // Spin::Up = +1 block 0, Spin::Down = -1 block 1
        const double x = sgn * spin * s[e < N ? e : e - N];                   // exponent is x * alpha, x = +-1
        const double f = x > 0.0 ? ep : em;
        cplx v = A[(size_t)j * lda + i];
        A[(size_t)j * lda + i] = make_double2(v.x * f, v.y * f);
    }
}
void launch_hubbard_vscale(const Launch& lc, const DevModel& hm, int side, int inverse, int k, cplx* A, int lda) {
    const size_t total = (size_t)hm.ng * hm.ng;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_hubbard_vscale, dim3(blocks, 1, lc.nb), dim3(256), 0, lc.st, hm, A, lda, side == DQMC_RIGHT ? 1 : 0,
                       inverse ? -1.0 : +1.0, k, lc.cs);
}

// e_m2a = exp(-2 alpha), e_p2a = exp(+2 alpha), computed by the host with the C library like the reference does
__global__ __launch_bounds__(256) void k_hubbard_slice(DevModel dm, DevUpdateState* __restrict__ us, const double* __restrict__ uniforms,
                                                        cplx* __restrict__ G, int k, double e_m2a, double e_p2a, size_t cs) {
    extern __shared__ double hsm[];                 // [2][N] columns G(:, site), [2][N] rows delta(site, :) - G(site, :)
    __shared__ int s_site, s_acc;
    __shared__ double s_fac[2];
    dm = chain_model(dm, cs); CHAIN(us); CHAIN(uniforms); CHAIN(G);
    const int N = dm.N, ng = dm.ng, tid = threadIdx.x;
    double* col = hsm;
    double* row = hsm + 2 * N;
    double* field = dm.phi + (size_t)k * N;
    unsigned long long cursor = 0, avail = 0;
    int accepted = 0;
    if (tid == 0) { cursor = us->pub.rng_consumed; avail = us->pub.rng_avail; }
    for (int count = 0; count < N; ++count) {
        if (tid == 0) {
            int site = -1, acc = 0;
            if (cursor < avail) {
                const double u = uniforms[cursor++];
                site = (int)(((double)(N - 1) - 0.0 + 1.0) * u);             // randInt(0, N-1): low + int((high - low + 1.0) * rand01())
                const double a = field[site];
                const double eu = a > 0.0 ? e_m2a : e_p2a;                    // exp(-2 alpha a)
                const double ed = a > 0.0 ? e_p2a : e_m2a;                    // exp(+2 alpha a)
                const double gu = G[(size_t)site * ng + site].x, gd = G[(size_t)(N + site) * ng + (N + site)].x;
                const double du = eu - 1.0, dd = ed - 1.0;
                const double ru = __dadd_rn(1.0, __dmul_rn(du, 1.0 - gu));    // weightRatioSingleFlip, no fma contraction
                const double rd = __dadd_rn(1.0, __dmul_rn(dd, 1.0 - gd));
                const double ratio = __dmul_rn(ru, rd);
                acc = ratio > 1.0;
                if (!acc) {
                    if (cursor < avail) acc = uniforms[cursor++] < ratio;
                    else site = -1;
                }
                if (site >= 0 && acc) {
                    s_fac[0] = du / ru;                                        // deltaSite / (1 + deltaSite (1 - g_ss))
                    s_fac[1] = dd / rd;
                    field[site] = -a;
                }
            }
            s_site = site; s_acc = acc;
        }
        __syncthreads();
        const int site = s_site, acc = s_acc;
        if (site < 0) break;                                                   // window ran dry (uniform decision)
        if (acc) {
            accepted += 1;
            for (int i = tid; i < N; i += 256) {
                col[i] = G[(size_t)site * ng + i].x;
                col[N + i] = G[(size_t)(N + site) * ng + (N + i)].x;
                row[i] = (i == site ? 1.0 : 0.0) - G[(size_t)i * ng + site].x;
                row[N + i] = (i == site ? 1.0 : 0.0) - G[(size_t)(N + i) * ng + (N + site)].x;
            }
            __syncthreads();
            const double f0 = s_fac[0], f1 = s_fac[1];
            for (int idx = tid; idx < N * N; idx += 256) {
                const int x = idx % N, y = idx / N;
                G[(size_t)y * ng + x].x -= __dmul_rn(__dmul_rn(col[x], f0), row[y]);
                G[(size_t)(N + y) * ng + (N + x)].x -= __dmul_rn(__dmul_rn(col[N + x], f1), row[N + y]);
            }
        }
        __syncthreads();                                                       // s_site / col / row are rewritten next round
    }
    if (tid == 0) {
        if (s_site < 0) us->pub.error = DQMC_ERNG;
        us->pub.rng_consumed = cursor;
        us->pub.lastAccRatio = (double)accepted / (double)N;
        us->updates_accepted += (unsigned long long)accepted;
    }
}
void launch_hubbard_slice(const Launch& lc, const DevModel& hm, DevUpdateState* us, const double* uniforms, cplx* G, int k,
                          double e_m2a, double e_p2a) {
    hipLaunchKernelGGL(k_hubbard_slice, dim3(1, 1, lc.nb), dim3(256), 4 * (size_t)hm.N * sizeof(double), lc.st, hm, us, uniforms, G, k,
                       e_m2a, e_p2a, lc.cs);
}

// acc: [0] sum G_ii up, [1] sum G_ii down, [2] sum G_<ij> up, [3] down, [4] sum G_ii,up G_ii,dn, [5] slices, [6 .. 6+N) zcorr
__global__ __launch_bounds__(256) void k_hubbard_measure(DevModel dm, const cplx* __restrict__ G, double* __restrict__ acc, size_t cs) {
    __shared__ double red[5][256];
    dm = chain_model(dm, cs); CHAIN(G); CHAIN(acc);
    const int N = dm.N, ng = dm.ng, tid = threadIdx.x;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int site = tid; site < N; site += 256) {
        const double gu = G[(size_t)site * ng + site].x, gd = G[(size_t)(N + site) * ng + (N + site)].x;
        v[0] += gu; v[1] += gd; v[4] += gu * gd;
        for (int dir = 0; dir < 4; ++dir) {
            const int nb = dm.neigh[dir * N + site];
            v[2] += G[(size_t)nb * ng + site].x;                                // gUp(site, site_neigh)
            v[3] += G[(size_t)(N + nb) * ng + (N + site)].x;
        }
    }
    for (int q = 0; q < 5; ++q) red[q][tid] = v[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int q = 0; q < 5; ++q) red[q][tid] += red[q][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        for (int q = 0; q < 5; ++q) acc[q] += red[q][0];
        acc[5] += 1.0;
    }
    const double u0 = G[0].x, d0 = G[(size_t)N * ng + N].x;
    for (int j = tid; j < N; j += 256) {
        double z;
        if (j == 0) z = -2.0 * u0 * d0 + u0 + d0;
        else {
            const double u0j = G[(size_t)j * ng].x, d0j = G[(size_t)(N + j) * ng + N].x;
            const double ujj = G[(size_t)j * ng + j].x, djj = G[(size_t)(N + j) * ng + (N + j)].x;
            z = u0 * ujj - u0 * djj + d0 * djj - d0 * ujj - u0j * u0j - d0j * d0j;
        }
        acc[6 + j] += z;
    }
}
void launch_hubbard_measure(const Launch& lc, const DevModel& hm, const cplx* G, double* acc) {
    hipLaunchKernelGGL(k_hubbard_measure, dim3(1, 1, lc.nb), dim3(256), 0, lc.st, hm, G, acc, lc.cs);
}
