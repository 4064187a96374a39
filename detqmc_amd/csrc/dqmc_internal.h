// Internal declarations shared by the HIP kernels and the C-ABI implementation.
// Not part of the public boundary (that is include/dqmc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dqmc_hip.h"

#include <stdlib.h>
// Developer A/B switches of individual kernels (tile shapes, 3M vs 4M products, ...).  None changes a result.  They exist only in
// builds with -DDQMC_DEV_KNOBS (DQMC_BUILD_DEFINES=-DDQMC_DEV_KNOBS python -m detqmc_amd.build --force); the shipped library reads
// ONE environment variable, DQMC_SYNC_CHECK (debug mode, dqmc_context.hip) -- everything else that selects an execution variant is
// a create-time parameter (dqmc_tuning, include/dqmc_hip.h).
static inline const char* dev_knob(const char* name) {
#ifdef DQMC_DEV_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

typedef double2 cplx;   // .x = re, .y = im; same bytes as dqmc_cplx / std::complex<double>

#define DQMC_MAX_MSF 4
#define DQMC_MAX_WDIM 64          // MSF * delaySteps <= 64 (W lives in LDS in the decision kernel)
#define DQMC_DECIDE_WIDE_MAX_CHAINS 32   // contexts of at most this many chains launch k_update_decide with 512 threads (kernels_update.hip)

// ---- batched chains ---------------------------------------------------------------------------
// One context can run nb independent Markov chains (replicas) in lockstep.  Every per-chain device buffer of
// chain b lives at (address of chain 0's buffer) + b * cs: all chains share one arena layout.  Kernels are
// launched with gridDim.z = nb and shift their per-chain pointer arguments by blockIdx.z * cs; read-only
// tables (plaquette tables, neighbours, tournament schedule) are shared and never shifted.
// sub: optional timing hooks for launches that belong to a kernel family of their own INSIDE a larger scope (profiling only): the
// trailing updates of the LU factorisation on the flush kernel (sub-family 0) and the products inside factorisations / triangular
// solves on k_zgemm (sub-family 1) -- both fill the GPU and are priced against a roof, unlike the panel kernels around them
enum { SUBFAM_LU_UPDATE = 0, SUBFAM_FACT_GEMM = 1, SUBFAM_COUNT = 4 };
struct SubProf { void (*begin)(void* user, int sub); void (*end)(void* user, int sub, double flops, double bytes); void* user; };
struct Launch { hipStream_t st; int nb; size_t cs; const SubProf* sub = nullptr; };
template<class T> __device__ __forceinline__ T* chain_ptr(T* p, size_t cs) {
    return p ? (T*)((char*)p + (size_t)blockIdx.z * cs) : p;
}
#define CHAIN(p) p = chain_ptr(p, cs)
// streaming (nontemporal) access to a matrix element: for operands a kernel touches exactly once per launch, so that they do not
// evict the panels / tables the same kernel re-reads from the L2 (one global_load/store_dwordx4 with the nt bit)
__device__ __forceinline__ cplx nt_load(const cplx* p) {
    cplx t;
    t.x = __builtin_nontemporal_load(&p->x); t.y = __builtin_nontemporal_load(&p->y);
    return t;
}
__device__ __forceinline__ void nt_store(cplx* p, const cplx& v) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
}
// XCD-aware launch shape for the wide MFMA kernels: consecutive workgroup ids go round-robin over the 8 XCDs (each with
// its own L2), so with a 1-D grid and   xcd = id % 8,  chain = 8 * (id / 8 / tiles) + xcd,  tile = (id / 8) % tiles
// all tiles of one chain run on ONE XCD and share the operand panels in that L2 instead of fetching them 8 times.
// Used when the number of chains is a multiple of 8; otherwise grid.z = chain as everywhere else.
template<class T> __device__ __forceinline__ T* chain_ptr_i(T* p, size_t cs, int chain) {
    return p ? (T*)((char*)p + (size_t)chain * cs) : p;
}
__device__ __forceinline__ void xcd_chain_tile(int tiles, int nb, int& chain, int& tile) {
    if (gridDim.z == 1 && nb > 1) {
        const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
        chain = (t / tiles) * 8 + xcd;
        tile = t % tiles;
    } else {
        chain = blockIdx.z;
        tile = blockIdx.x;
    }
}

// Everything a kernel needs to know about the model; lives in device memory, one per context.
struct DevModel {
    int opdim, MSF, L, N, ng, m, s, n, D, P;   // P = plaquettes per subgroup = N/4
    int phi2bosons;
    int decide_nt;     // threads per workgroup of k_update_decide: 0 automatic, 256, 512 (dqmc_tuning::decide_threads)
    int pbudget;       // proposals per delayed-update block (0: no limit); a launch-balancing knob, the chain does not depend on it
    int dbg;           // bit 3: phase timers of k_update_decide; only ever set in builds with -DDQMC_DECIDE_TIMING (always 0 otherwise)
    int dense;         // CB_NONE: the hopping part is a dense GEMM done by the host loop, the chain kernel
                       // only applies e^{+-dtau V}; ov/ovinv are 1 (mu sits inside propK)
    double dtau, r, c, u, lambda;   // r: chain 0's value at create time only -- kernels read DevUpdateState::r
    double ov[2];      // e^{+dtau mu_band}   (detsdwopdim.cpp:2037-2038)
    double ovinv[2];   // e^{-dtau mu_band}   (detsdwopdim.cpp:2137-2138)
    // plaquette sites [sub][4][P] (sub 0: even corners, sub 1: odd corners; detsdwopdim.cpp:1776-1785).  All
    // plaquette tables are structure-of-arrays over the plaquette index: lanes work on consecutive plaquettes
    const int* psites;
    // 4x4 complex plaquette exponentials [band][signIdx][sub][16 (row-major entry)][P]; sub 1 holds the
    // half-step matrices, sub 0 the full-step ones (symmetric break-up, detsdwopdim.cpp:1846-1865);
    // signIdx 0: e^{-dtau K}, 1: e^{+dtau K}.  Used with a magnetic flux (complex Hermitian matrices).
    const cplx* pmats;
    // without flux the matrices are real with four distinct entries, rows (a b c d)(b a d c)(c d a b)(d c b a)
    // (cb_assaad_applyBondFactorsLeft, detsdwopdim.cpp:1688-1756): [band][signIdx][sub][4][P]
    const double* pabcd;
    int pm_real;
    // the same two tables with HALF steps for both subgroups (shiftGreenSymmetric, detsdwopdim.cpp:4507-4563)
    const cplx* pmats_h;
    const double* pabcd_h;
    // fields
    double* phi;       // [m+1][opdim][N]
    double* coshT;     // [m+1][N]
    double* sinhT;     // [m+1][N]
    // cdwU != 0 (cdw_on): the discrete four-valued field l_i(tau_k) in {+-1, +-2} (kept as doubles: it travels with the other per-site
    // scalars of a proposal) and its caches cosh / sinh(sqrt(dtau) cdwU eta(l)) (coshTermCDWl / sinhTermCDWl, detsdwopdim.cpp:1138-1163)
    int cdw_on;
    double* cdwl;      // [m+1][N]
    double* cdwC;      // [m+1][N]
    double* cdwS;      // [m+1][N]
    double cdw_cosh[2], cdw_sinh[2];   // |l| = 1, 2 (sinh for l > 0)
    double cdw_gamma[2];               // cdwl_gamma(|l|) (detsdwopdim.h:1209-1220)
    const int* neigh;  // [4][N]  XPLUS, XMINUS, YPLUS, YMINUS (neighbortable.h:34-36)
    // Hubbard replica (dqmc_params::model == DQMC_MODEL_HUBBARD): phi holds the Ising auxiliary field (+-1.0, opdim = 1),
    // the hopping part is the dense propagator (dense = 1), the site-diagonal part is e^{+-alpha s} (kernels_hubbard.hip)
    int hubbard;
    double hub_exp_alpha[2];   // e^{+alpha}, e^{-alpha}, cosh(alpha) = e^{dtau U / 2} (dethubbard.cpp:55)
};
__device__ __forceinline__ DevModel chain_model(DevModel dm, size_t cs) {
    dm.phi = chain_ptr(dm.phi, cs); dm.coshT = chain_ptr(dm.coshT, cs); dm.sinhT = chain_ptr(dm.sinhT, cs);
    if (dm.cdw_on) { dm.cdwl = chain_ptr(dm.cdwl, cs); dm.cdwC = chain_ptr(dm.cdwC, cs); dm.cdwS = chain_ptr(dm.cdwS, cs); }
    return dm;
}

struct DevUpdateState {
    dqmc_update_state pub;   // mirrored to the host on request
    int site_cursor;         // next site to be proposed in the current slice
    int acc_count;           // accepted proposals in the current slice
    int block_j;             // accepted updates in the block the last decision launch produced
    int slice_done;
    int flush_k;             // K = MSF j of the block whose flush is in flight (pipelined update: block_j already belongs to the next block)
    int nd_has;              // rotate / scale proposals: the Box-Muller stack of NormalDistribution holds a value (normaldistribution.h), ...
    double nd_cached;        // ... this one; reset at the top of every updateInSlice
    int chol_fail;           // set by k_chol64 when a Cholesky-QR panel fails its pivot test; read and cleared by the host after the factorisation
    double r;                // this chain's exchange parameter (differs between the chains of a batch)
    int block_sites[DQMC_MAX_WDIM];
    unsigned long long blocks_nonempty;  // delayed-update blocks that accepted at least one update (-> real flushes)
    unsigned long long updates_accepted; // accepted local updates (sum of the block ranks j); must follow blocks_nonempty
    unsigned long long dbg_cycles[16];   // developer phase timers of the decision kernel (-DDQMC_DECIDE_TIMING builds only)
};

// ---- launchers (implemented in the kernels_*.hip files) ---------------------------------------
// checkerboard chain: A <- prod B_k A etc. for k = kfirst, kfirst+kstep, ... (count slices)
void launch_bmult(const Launch& lc, const DevModel* dm, const DevModel& hm, int side, int inverse,
                  int kfirst, int kstep, int kcount, cplx* A, int lda, int shift = 0);   // shift: half-step hopping passes only

// C = alpha-less complex GEMM on MFMA f64: C[MxN] (+)= op(A)[MxK] . diag(kscale) . op(B)[KxN]
struct GemmArgs {
    const cplx* A; int lda; int opA;      // op: 0 = N, 1 = conjugate transpose
    const cplx* B; int ldb; int opB;
    cplx* C; int ldc;
    int M, N, K;
    const int* Kdev;            // if non-null: K = min(K, *Kdev * Kmul) read on the device
    int Kmul;
    const double* kscale;       // optional scale of the contraction index
    int kscale_invert;          // use 1/kscale
    const double* rowscale;     // optional epilogue: acc *= rowscale[i] * colscale[j]
    const double* colscale;
    int accumulate;             // C += instead of C =
    int negate;                 // the product enters with a minus sign (C -= A B with accumulate)
    int sharedA, sharedB;       // operand is one matrix for all chains (not shifted by the chain stride)
    const int* a_kgather;       // opA == 0 only: column k of op(A) is column a_kgather[k] of A
    int b_lower;                // op(B) is lower triangular (entries with k < j are zero): the k loop of a column tile starts at its first column
    // split-K (skinny products with a long contraction index, e.g. Q_prev^H A_j of the block Gram-Schmidt QR: few output tiles, K = n):
    // ksplit > 1 slices of K are computed by separate workgroups into part[slice][N][M] and summed in a fixed order by a second kernel
    int ksplit; cplx* part;
    size_t part_count;          // capacity of part in complex numbers (0: eight slices of M x N, the round-3 contract)
    int tag;                    // 1: a product inside a factorisation (LU trailing update, triangular solve) -- same code, its own kernel
                                // name (template argument), so that profiles keep it apart from the model's n_g^3 products
};
void launch_gemm(const Launch& lc, const GemmArgs& a);
// G += X GrT^T, K = min(Kmax, *Kdev * Kmul): the delayed-update flush as a register-only read-modify-write stream.  X and GrT are
// n x K8 (K8 = K rounded up to a multiple of 8, columns K .. K8 - 1 zero), both with leading dimension ld
void launch_flush(const Launch& lc, const cplx* X, const cplx* GrT, int ld, cplx* G, int ldc, int n, int Kmax,
                  const int* Kdev, int Kmul, int tag = 0);    // G (n x n) += X GrT^T, K = min(Kmax, *Kdev * Kmul) (Kdev may be null); tag 1: own kernel name

// one-sided Jacobi SVD, M = U diag(d) V^H, d descending.  work: A (n*n), V (n*n), norms(n), rank(n),
// flag (int).  Host-driven sweep loop with one flag read-back per sweep.  Returns sweeps used or <0.
struct SvdWork {
    cplx* A; cplx* V; double* norms; double* rnorms; int* rank; int* flagT;
    unsigned long long* flag; double* last_residual;
    unsigned long long* hflag;       // mapped pinned host memory: [0] sequence number, [1] residual bits
    unsigned long long* hslot_dev;   // device alias of hflag
    unsigned long long* seqctr;      // device-side publish counter
    unsigned long long* host_seq;    // host-side expectation
    const int* rounds; int nrounds; int nblk;   // tournament table [nrounds][nblk/2][2]
    hipGraphExec_t* sweep_graph;                // lazily captured: memset + all rounds of one Jacobi sweep
};
// optional timing hooks around each batch of back-to-back round launches (one Jacobi sweep)
struct SvdProfHooks { void (*begin)(void* user); void (*end)(void* user, int launches); void* user; };
int run_svd(const Launch& lc, int n, const cplx* M, int ldm, const double* colscale, const double* rowscale,
            cplx* U, double* d, cplx* Vt, const SvdWork& w, int max_sweeps, const SvdProfHooks* hooks = nullptr);
int svd_block_cols(int n);      // columns per block used by the Jacobi kernel for this n

// local updates
// diagonal entries and off-diagonal scale of e^{sign dtau V} at one site: (0,0) = (2,2) = c0, (1,1) = (3,3) = c1, off-diagonal entries
// carry xs (evMatrix, detsdwopdim.cpp:3187-3229; cd / cmd of :2003-2006).  idx = k N + site.
__device__ __forceinline__ void cdw_site_terms(const DevModel& dm, size_t idx, double sign, double c, double& c0, double& c1, double& xs) {
    if (dm.cdw_on) {
        const double cC = dm.cdwC[idx], sC = dm.cdwS[idx];
        c0 = c * cC - sign * sC; c1 = c * cC + sign * sC; xs *= cC;
    } else { c0 = c; c1 = c; }
}
void launch_cdw_terms(const Launch& lc, const DevModel& hm);      // cdwC / cdwS from cdwl, all slices
// cdw_mode 0: phi proposals (with the cdw terms in e^{dtau V} when cdw_on); 1: the cdwl pass (proposeNewCDWl, :4173-4182)
void launch_update_decide(const Launch& lc, const DevModel* dm, const DevModel& hm, DevUpdateState* us,
                          const double* uniforms, const cplx* G, cplx* W, int k, int first, int thermal, int cdw_pass = 0,
                          const cplx* Gwin = nullptr, int winP = 0,       // winP > 0: G entries from the window copy (k_update_window)
                          int proposal = 0, int adapt_what = 0, int reset_nd = 0);   // proposal: DQMC_PROPOSE_*; adapt_what: 0 box, 1 rotate, 2 scale (+ 4: adaptScaleVariance)
void launch_update_window(const Launch& lc, const DevModel& hm, DevUpdateState* us, const cplx* G, const cplx* X, const cplx* GrT,
                          cplx* Gw, int P);
void launch_update_gather(const Launch& lc, const DevModel& hm, const DevUpdateState* us, const cplx* G,
                          const cplx* W, cplx* X, cplx* GrT);

// Hubbard replica (kernels_hubbard.hip)
void launch_hubbard_vscale(const Launch& lc, const DevModel& hm, int side, int inverse, int k, cplx* A, int lda);
void launch_hubbard_slice(const Launch& lc, const DevModel& hm, DevUpdateState* us, const double* uniforms, cplx* G, int k,
                          double e_m2a, double e_p2a);
void launch_hubbard_measure(const Launch& lc, const DevModel& hm, const cplx* G, double* acc);

// misc elementwise
void launch_cosh_sinh(const Launch& lc, const DevModel& hm);
void launch_set_identity(const Launch& lc, cplx* A, int n);
void launch_conj_transpose(const Launch& lc, const cplx* A, cplx* B, int n);
void launch_add_diag(const Launch& lc, cplx* A, const double* d, int n);
void launch_copy(const Launch& lc, const cplx* A, cplx* B, size_t count);
void launch_copy_bytes(const Launch& lc, const void* src, void* dst, size_t bytes);   // same buffer of every chain
void launch_phi_sq_sum(const Launch& lc, const DevModel& hm, double* out);
void launch_gather_scalars(const Launch& lc, const double* src, double factor, double* dst);      // dst[b] = factor * src of chain b (dst: plain device array)
void launch_phi_action(const Launch& lc, const DevModel& hm, const DevUpdateState* us, double* out);
void launch_phi_shift(const Launch& lc, const DevModel& hm, const double* shifts /* shared buffer [nchains][opdim] */);
// fermionic observables of one time slice accumulated from the shifted Green's function gs (kernels_measure.hip)
// acc layout (doubles): [0] sum Re gs, [1] Re tr gs, [2] occDiffSq sum, [3] slices, then pairPlus[N], pairMinus[N],
// SX[(2L-1)^2] (re, im), SY[(2L-1)^2] (re, im)
void launch_measure_accum(const Launch& lc, const DevModel& hm, const cplx* gs, double* acc);
size_t measure_accum_doubles(int N, int L);

// ---- QR / UDT building blocks (kernels_qr.hip) ------------------------------------------------
struct SvdProfHooks;
struct QrWork { cplx* V; cplx* T; cplx* W; cplx* W2; cplx* Rneg; const SvdProfHooks* apply_hooks; cplx* part; size_t part_count; int* err; };   // part: split-K scratch (n > 1024); err: per-chain failure flag of the Cholesky-QR panels (DevUpdateState::chol_fail)   // hooks: optional timing of the k_qr_apply launches
int run_qr(const Launch& lc, int n, cplx* A, cplx* Q, const QrWork& w);                 // A -> R in place, Q explicit (Q == nullptr: reflectors only)
int run_qr_apply_q(const Launch& lc, int n, cplx* C, const QrWork& w, int trans);
// large n: QR by block Gram-Schmidt with reorthogonalisation + Cholesky-QR2 panels, all on the GEMM kernel (kernels_qr.hip); Q is always explicit
int run_qr_bgs(const Launch& lc, int n, cplx* A, cplx* Q, const QrWork& w);
void qr_reset_workspace(const Launch& lc, int n, const QrWork& w);      // zero w.V / w.T before Householder panels follow a block Gram-Schmidt run
int run_trsm_right_upper(const Launch& lc, int n, const cplx* R, cplx* C, const QrWork& w, int trans = 0, int unit = 0);   // C <- C R^-1 (trans: R = (stored lower triangle)^H; unit: unit diagonal)
#define LU_SWAP_INTS 128
int run_lu(const Launch& lc, int n, cplx* A, int* perm, int* swaps, cplx* tneg);    // tneg: scratch of n * 32 complex per chain                                 // P A = L U in place (n <= 512, else -1), kernels_lu.hip
void launch_gather_scale_cols(const Launch& lc, const cplx* X, const double* cs, const int* perm, int n, cplx* Y);   // Y[:, j] = X[:, perm[j]] cs[perm[j]]
void launch_udt_init(const Launch& lc, const cplx* M, int ldm, const double* cs, const double* rs, const int* perm,
                     int transpose, cplx* W, int n);
void launch_udt_diag(const Launch& lc, const cplx* R, int n, double* d);
void launch_udt_lazy(const Launch& lc, const double* d, const int* perm, int n, double* dinv, int* perm_inv);   // 1/d and the inverse permutation
void launch_udt_tmat(const Launch& lc, const cplx* R, const double* d, const int* perm, int n, cplx* Tt);
void launch_permute_scale_cols(const Launch& lc, const cplx* X, const double* cs, const int* perm, int n, cplx* Y);
void launch_split_scales(const Launch& lc, const double* d, int n, double* dmax_inv, double* dmin);
void launch_logdet_vector(const Launch& lc, const cplx* R, const double* a, const double* b, int n, double* sv);
// column norms / ranks shared with the SVD path (kernels_svd.hip)
void launch_scaled_norms_rank(const Launch& lc, const cplx* M, int ldm, const double* cs, const double* rs, int transpose,
                              int n, double* norms, int* rank, double* scratch_d);
