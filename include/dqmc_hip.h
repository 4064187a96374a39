/*
 * dqmc_hip.h -- C ABI of the MI355X (gfx950) DQMC sweep kernels.
 *
 * This is the drop-in boundary for the reference's DetModelGC/DetSDW hot path
 * (crstnbr/detqmc).  The reference has NO runtime plugin boundary for this path: DetQMC<Model> is
 * a class template and DetModelGC::sweep_skeleton takes the B-multiply / update routines as
 * template callables (src/detqmc.h:58-59, src/detmodel.h:250-265).  Each entry point below
 * replaces one of those callables / member functions; the reference interface it replaces is
 * cited as file:line relative to /root/reference/src.  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - complex fp64, interleaved (re, im) == std::complex<double> == arma::cx_double
 *   - matrices are n_g x n_g, COLUMN-major, leading dimension n_g, n_g = MSF*N,
 *     MSF = (opdim == 3 ? 4 : 2) (src/detsdwopdim.h:161), N = L*L, site = y*L + x
 *   - phi is laid out like the reference's arma::Cube phi(N, OPDIM, m+1): index
 *     site + N*(dim + OPDIM*k); slice k = 0 is unused (src/detsdwopdim.h:461-466)
 *   - every function returns 0 on success, a negative DQMC_E* code otherwise; nothing throws
 *     across the boundary; dqmc_last_error() gives the text
 *   - one dqmc_ctx == one replica, bound to one device and one HIP stream; not thread-safe;
 *     different contexts may be driven from different host threads
 *   - functions whose name ends in _host take/return caller-owned HOST buffers and synchronise;
 *     all others only enqueue work on the context's stream
 */
#ifndef DQMC_HIP_H_
#define DQMC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dqmc_ctx dqmc_ctx;

typedef struct dqmc_cplx { double re, im; } dqmc_cplx;

enum {
    DQMC_OK = 0,
    DQMC_EINVAL = -1,     /* bad argument / unsupported parameter (reference: ParameterWrong) */
    DQMC_EHIP = -2,       /* HIP runtime error */
    DQMC_ENOCONV = -3,    /* decomposition did not converge (reference: "SVD failed (std)", udv.h:77-88) */
    DQMC_ERNG = -4,       /* device ran out of pre-drawn uniforms */
    DQMC_ENODEV = -5      /* no usable GPU */
};

enum { DQMC_BC_PBC = 0, DQMC_BC_APBC_X = 1, DQMC_BC_APBC_Y = 2, DQMC_BC_APBC_XY = 3 };
enum { DQMC_LEFT = 0, DQMC_RIGHT = 1 };
enum { DQMC_UP = +1, DQMC_DOWN = -1 };
enum { DQMC_STAB_SVD = 0, DQMC_STAB_QR = 1 };
enum { DQMC_MODEL_SDW = 0, DQMC_MODEL_HUBBARD = 1 };

/* Execution choices that change NO result (the parity tests hold for every value); 0 = automatic everywhere.  They are
 * create-time parameters of a context: nothing about a context's launch schedule depends on the environment or on other
 * contexts of the process. */
typedef struct dqmc_tuning {
    int32_t pipeline;          /* delayed updates: the flush of block b on a second stream next to the decisions of block b + 1
                                  (which read a compact, already updated copy of their proposal window).  0: automatic (n_g > 1024
                                  and at least two chains), 1: on, -1: off.  dqmc_get_schedule_info tells what ran */
    int32_t qr_variant;        /* QR mode, chain factorisations: 0 automatic (Householder panels up to n_g = 1024, block
                                  Gram-Schmidt + Cholesky-QR2 above), 1: Householder, 2: block Gram-Schmidt */
    int32_t green_variant;     /* QR mode, inverse inside greenFromUdV: 0 automatic (LU with partial pivoting for n_g <= 512,
                                  QR above), 1: QR */
    int32_t max_jacobi_sweeps; /* SVD mode: sweep budget of the one-sided Jacobi SVD, 0 = 80; exhausting it is DQMC_ENOCONV
                                  ("SVD failed", udv.h:77-88) */
    int32_t proposal_budget;   /* proposals per delayed-update block: 0 automatic (2 delaySteps for delaySteps >= 8), -1 no
                                  limit, > 0 that many (at least delaySteps) */
    int32_t decide_threads;    /* threads per workgroup of the decision kernel: 0 automatic (512 for O(1) / O(2) contexts of at most 32
                                  chains, else 256), 256, 512 (O(3): always 256).  Launch shape only: the chain does not depend on it */
    int32_t reserved[2];
} dqmc_tuning;

/* ModelParamsDetSDW fields the kernels depend on (src/detsdwparams.h:24-120) */
typedef struct dqmc_params {
    int32_t opdim;        /* 1, 2 or 3 */
    int32_t L;            /* even */
    int32_t m;            /* time slices */
    int32_t s;            /* stabilisation interval, s < m */
    int32_t delaySteps;   /* D, 1..N, MSF*D <= 64 */
    int32_t bc;           /* DQMC_BC_* */
    int32_t weakZflux;    /* only with opdim == 2 */
    int32_t phi2bosons;
    int32_t device;       /* HIP device ordinal */
    int32_t stabilisation; /* DQMC_STAB_SVD (reference-exact UdV = SVD) or DQMC_STAB_QR (pre-pivoted Householder UDT) */
    int32_t cb_none;      /* 0: checkerboard break-up CB_ASSAAD_BERG (every shipped config); 1: checkerboard=false,
                             dense B_k = e^{-dtau V_k} e^{-dtau K} (computeBmatSDW, detsdwopdim.cpp:1309-1485) */
    int32_t model;        /* DQMC_MODEL_SDW (default) or DQMC_MODEL_HUBBARD: the reference's DetHubbard (src/dethubbard.{h,cpp}), both
                             spin sectors in one block-diagonal 2N x 2N matrix.  Hubbard reads: L, m, s, dtau, txhor = t, u = U,
                             mux = mu, cb_none = !checkerboard (propagator e^{-dtau T} by diagonalisation, or the checkerboard
                             product form of dethubbard.cpp:717-770), stabilisation, device; opdim must be 1 (the field is the
                             Ising auxiliary field, phi = +-1), delaySteps 1 */
    double dtau, r, c, u, lambda;
    double txhor, txver, tyhor, tyver;
    double mux, muy;
    double accRatio;      /* target acceptance for the box step adaptation */
    double cdwU;          /* != 0: the discrete field l_i(tau) in {+-1, +-2} next to phi (detsdwparams.h:61; evMatrix,
                             detsdwopdim.cpp:3187-3229); dqmc_update_slice then runs the cdwl pass behind the phi pass (:2474-2485) */
    int32_t rng_window_per_site;   /* capacity of the window of pre-drawn uniforms, per site and time slice of a sweep; 0 = what box proposals
                                      with repeatUpdateInSlice = 1 can consume at most (opdim + 1, + 2 with cdwU).  Callers that use
                                      dqmc_update_slice_ex with more passes or the rotate / scale proposals (whose Gaussian draws consume
                                      a variable number) size it: repeat x (opdim + 1) resp. repeat x 8 (+ 2 with cdwU) */
    int32_t reserved_model;
    dqmc_tuning tuning;   /* all zero = automatic */
} dqmc_params;

/* AdjustmentData + slice bookkeeping that lives on the device between calls
 * (src/detsdwopdim.h:481-577, RunningAverage.h) */
typedef struct dqmc_update_state {
    double phiDelta;
    double targetAccRatio;
    double lastAccRatio;
    double ra_runningAverage;
    double ra_values[100];
    int32_t ra_samplesAdded;
    int32_t ra_head;
    uint64_t rng_consumed;   /* uniforms consumed from the pushed window */
    uint64_t rng_avail;      /* size of the pushed window */
    int32_t error;           /* DQMC_ERNG if the window ran dry */
    int32_t reserved;
    /* rotate / scale proposals of the O(3) model (spinProposalMethod != box): AdjustmentData::angleDelta (the minimal cos(theta) of a
     * rotation), scaleDelta (width of the Gaussian |phi|^3 update), the bisection bounds of their adaptation and the running
     * averages accRatioLocal_rotate_RA / _scale_RA (src/detsdwopdim.h:489-530, src/detsdwopdim.cpp:3299-3375) */
    double angleDelta, scaleDelta;
    double curminAngleDelta, curmaxAngleDelta, curminScaleDelta, curmaxScaleDelta;
    double rot_runningAverage;
    double rot_values[100];
    double scl_runningAverage;
    double scl_values[100];
    int32_t rot_samplesAdded, rot_head, scl_samplesAdded, scl_head;
} dqmc_update_state;

/* ---- lifetime ------------------------------------------------------------------------- */
/* replaces DetSDW ctor set-up of hopping constants / 4-site exponentials
 * (detsdwopdim.cpp:217-264, :1598-1684) */
int dqmc_create(const dqmc_params* p, dqmc_ctx** out);
/* Batched replicas: ONE context that advances `nchains` independent Markov chains (the replicas of a
 * parallel-tempering run, src/detqmcpt.h: one replica per MPI rank there) in lockstep -- every kernel launch
 * carries all chains (grid.z = chain), which is what fills the 256 CUs at the lattice sizes of interest.
 * All chains share the lattice/temperature parameters of *p; fields, RNG windows, update state and the
 * exchange parameter r are per chain.  The compute entry points (udv_setup, advance, wrap, update_slice,
 * backup) act on all chains; the host-buffer entry points (set/get fields, Green's function, singular
 * values, UdV, uniforms, update state, exchange parameter / action, restore) act on the chain chosen with
 * dqmc_select_chain (default 0).  dqmc_create == dqmc_create_batch with nchains = 1. */
int dqmc_create_batch(const dqmc_params* p, int nchains, dqmc_ctx** out);
int dqmc_select_chain(dqmc_ctx* ctx, int chain);
int dqmc_num_chains(dqmc_ctx* ctx);
void dqmc_destroy(dqmc_ctx* ctx);
const char* dqmc_last_error(void);
int dqmc_synchronize(dqmc_ctx* ctx);
/* raw stream handle (hipStream_t) for event timing by the harness */
void* dqmc_stream(dqmc_ctx* ctx);

/* ---- fields (a22) ---------------------------------------------------------------------- */
/* upload phi and recompute cosh/sinh caches: updateCoshSinhTermsPhi (detsdwopdim.cpp:1175-1181) */
int dqmc_set_fields_host(dqmc_ctx* ctx, const double* phi);
int dqmc_get_fields_host(dqmc_ctx* ctx, double* phi, double* coshTermPhi, double* sinhTermPhi);
/* cdwU != 0: the discrete field of the selected chain, cdwl[k * N + site] in {+-1, +-2} (slice 0 unused); set recomputes
 * coshTermCDWl / sinhTermCDWl (updateCoshSinhTermsCDWl, detsdwopdim.cpp:1183-1190).  After dqmc_create the field is +1
 * everywhere (setupConstantField, :1116-1128).  DQMC_EINVAL on a context created with cdwU == 0. */
int dqmc_set_cdwl_host(dqmc_ctx* ctx, const int32_t* cdwl);
int dqmc_get_cdwl_host(dqmc_ctx* ctx, int32_t* cdwl);
/* all chains of a batched context in ONE transfer: phi_all = nchains cubes of (m+1) * opdim * N doubles, back to back */
int dqmc_set_fields_all_host(dqmc_ctx* ctx, const double* phi_all);
int dqmc_get_fields_all_host(dqmc_ctx* ctx, double* phi_all);

/* ---- checkerboard B-multiplies (a10-a14) -------------------------------------------------- */
/* A <- B(k2,k1) A | B(k2,k1)^-1 A | A B(k2,k1) | A B(k2,k1)^-1 on a HOST matrix:
 * checkerboard{Left,Right}MultiplyBmat[Inv] (detsdwopdim.cpp:2076-2090, 2172-2186, 2307-2324,
 * 2406-2420), i.e. the four callables of sweep_skeleton (detmodel.h:256-260). */
int dqmc_bmult_host(dqmc_ctx* ctx, int side, int inverse, int k2, int k1, dqmc_cplx* A);

/* ---- UdV decomposition and dense products (a2, L1) ---------------------------------------- */
/* udvDecompose (udv.h:68-102): M = U diag(d) V_t^H, d descending */
int dqmc_udv_decompose_host(dqmc_ctx* ctx, const dqmc_cplx* M, dqmc_cplx* U, double* d, dqmc_cplx* V_t,
                            int* sweeps_used);
/* C = op(A) op(B), op = identity (0) or conjugate transpose (1): the zgemm calls behind
 * detmodel.h:784-815 */
int dqmc_gemm_host(dqmc_ctx* ctx, int opA, int opB, const dqmc_cplx* A, const dqmc_cplx* B, dqmc_cplx* C);

/* ---- stabilised Green's function (a3-a8) ---------------------------------------------------- */
/* setupUdVStorage_and_calculateGreen_skeleton (detmodel.h:680-713): storage[0..n], G(beta) */
int dqmc_udv_setup(dqmc_ctx* ctx);
/* advanceUpGreen(l) / advanceDownGreen(l) incl. greenFromUdV / greenFromEye_and_UdV
 * (detmodel.h:1109-1163, 956-1017, 769-860) */
int dqmc_advance(dqmc_ctx* ctx, int dir, int l);
/* wrapUpGreen(k): G <- B_{k+1} G B_{k+1}^-1 ; wrapDownGreen(k): G <- B_k^-1 G B_k
 * (detmodel.h:1236-1259, 1066-1095) */
int dqmc_wrap(dqmc_ctx* ctx, int dir, int k);
/* sweepUp resets storage[0] to the identity (detmodel.h:1293-1295) */
int dqmc_reset_storage0(dqmc_ctx* ctx);

/* ---- local updates (a17-a20) ---------------------------------------------------------------- */
/* replace the window of pre-drawn uniforms (0,1) the device consumes in stream order */
int dqmc_push_uniforms_host(dqmc_ctx* ctx, const double* u, size_t n);
/* the same for ALL chains of a batched context in one transfer: u = nchains windows of n uniforms each, back to back */
int dqmc_push_uniforms_all_host(dqmc_ctx* ctx, const double* u, size_t n);
/* updateInSlice (detsdwopdim.cpp:2428-2489, delayed updates :3023-3175, box proposals :3922-3931,
 * deltaSPhi :4186-4239, get_delta_forsite :3179-3289); thermalization != 0 adds the step-size
 * adaptation of updateInSliceThermalization (:3294-3375) */
int dqmc_update_slice(dqmc_ctx* ctx, int k, int thermalization);
/* The same with the other proposal kinds and repeatUpdateInSlice (src/detsdwopdim.cpp:2438-2470): `proposal` = DQMC_PROPOSE_BOX
 * (proposeNewPhiBox), _ROTATE (proposeRotatedPhi, :3945-4002), _SCALE (proposeScaledPhi, :4016-4077), _ROTATE_AND_SCALE
 * (proposeRotatedScaledPhi, :4092-4170) -- the last three for opdim == 3 only, as in the reference; `repeat` passes over the slice
 * (each a full updateInSlice_delayed; lastAccRatio is the last pass's); with thermalization != 0 the running average and step
 * parameter named by `adapt` are updated once, from the last pass (updateInSliceThermalization, :3294-3375): DQMC_ADAPT_BOX (phiDelta),
 * _ROTATE (angleDelta), _SCALE (scaleDelta, only moved with adapt_scale_variance != 0).  Which kind a sweep uses (rotate_then_scale
 * alternates with performedSweeps, rotate_and_scale alternates the ADAPTED quantity every 100 sweeps) is the host layer's business. */
enum { DQMC_PROPOSE_BOX = 0, DQMC_PROPOSE_ROTATE = 1, DQMC_PROPOSE_SCALE = 2, DQMC_PROPOSE_ROTATE_AND_SCALE = 3 };
enum { DQMC_ADAPT_BOX = 0, DQMC_ADAPT_ROTATE = 1, DQMC_ADAPT_SCALE = 2 };
int dqmc_update_slice_ex(dqmc_ctx* ctx, int k, int thermalization, int proposal, int adapt, int adapt_scale_variance, int repeat);
/* Which schedule dqmc_update_slice runs for this context (latched at dqmc_create from dqmc_tuning::pipeline and the shape of the
 * context) and how many delayed-update blocks have gone through each since dqmc_create. */
typedef struct dqmc_schedule_info {
    int32_t pipelined;              /* 1: flush on the second stream next to the next block's decisions; 0: strictly sequential */
    int32_t proposal_budget;        /* proposals per block in effect (0: no limit) */
    uint64_t blocks_pipelined;      /* decide / gather / flush rounds launched in the pipelined form */
    uint64_t blocks_sequential;     /* ... in the sequential form */
    int32_t qr_block_gram_schmidt;  /* 1: chain factorisations by block Gram-Schmidt + CholQR2, 0: Householder panels */
    int32_t green_lu;               /* 1: inverse inside greenFromUdV by LU, 0: by QR */
    uint64_t cholqr_fallbacks;      /* factorisations in which a CholQR panel lost definiteness and was redone with Householder panels */
} dqmc_schedule_info;
int dqmc_get_schedule_info(dqmc_ctx* ctx, dqmc_schedule_info* out);
int dqmc_get_update_state_host(dqmc_ctx* ctx, dqmc_update_state* out);
int dqmc_get_update_states_all_host(dqmc_ctx* ctx, dqmc_update_state* out /* [nchains] */);
int dqmc_set_update_state_host(dqmc_ctx* ctx, const dqmc_update_state* in);

/* ---- state access ------------------------------------------------------------------------------ */
int dqmc_get_green_host(dqmc_ctx* ctx, dqmc_cplx* out);
int dqmc_set_green_host(dqmc_ctx* ctx, const dqmc_cplx* in, int currentTimeslice);
/* green_inv_sv (detmodel.h:466): singular values of G^-1 in SVD mode; in QR mode a positive vector with
 * the same log-sum (= log|det G^-1|), which is all the global moves use (detsdwopdim.cpp:3613-3620) */
int dqmc_get_sv_host(dqmc_ctx* ctx, double* out);
int dqmc_get_sv_all_host(dqmc_ctx* ctx, double* out /* [nchains][n_g] */);
int dqmc_get_udv_host(dqmc_ctx* ctx, int l, dqmc_cplx* U, double* d, dqmc_cplx* V_t);
int dqmc_current_timeslice(dqmc_ctx* ctx);

/* ---- global-move support (a21) ------------------------------------------------------------------ */
/* globalMoveStoreBackups / globalMoveRestoreBackups (detsdwopdim.cpp:3886-3917): swap G, sv,
 * UdV storage, copy fields */
int dqmc_backup(dqmc_ctx* ctx);
int dqmc_restore(dqmc_ctx* ctx);
/* global shift move on the device (attemptGlobalShiftMove, detsdwopdim.cpp:3565-3644): phiAction (:4242-4300) of every chain from
 * the resident field, out[nchains]; addGlobalRandomDisplacement (:3755-3763) of every chain, shifts[nchains][opdim] */
int dqmc_phi_action_all_host(dqmc_ctx* ctx, double* out);
int dqmc_shift_fields_all_host(dqmc_ctx* ctx, const double* shifts);
/* 1/2 dtau sum phi^2 (get_exchange_action_contribution, detsdwopdim.cpp:5205-5216) */
int dqmc_exchange_action_host(dqmc_ctx* ctx, double* out);
/* the same for every chain of the context, written to a caller-owned DEVICE array of nchains doubles (the send buffer of the
 * replica-exchange all_gather, src/detqmcpt.h:1003-1010, without a host hop); returns when the values are there */
int dqmc_exchange_actions_device(dqmc_ctx* ctx, double* out_dev);

/* ---- fermionic measurements --------------------------------------------------
 * shiftGreenSymmetric (src/detsdwopdim.cpp:4507-4612): e^{-dtau K/2} G e^{+dtau K/2} of the selected chain */
int dqmc_shift_green_symmetric_host(dqmc_ctx* ctx, dqmc_cplx* out);
/* initMeasurements / measure(k), G-dependent part (src/detsdwopdim.cpp:458-505, :545-899), all chains: the slice's
 * contributions are accumulated on the device.  Accumulator layout (doubles, dqmc_measure_accum_size of them):
 * [0] greenK0 sum, [1] greenLocal sum, [2] occDiffSq sum, [3] slices measured, pairPlus[N], pairMinus[N],
 * S_X[(2L-1)^2] and S_Y[(2L-1)^2] as (re, im): S_band(dx, dy) = sum over site pairs with r_i - r_j = (dx, dy) of
 * g_band,up(i,j) + g_band,down(i,j), bin index (dy + L-1) (2L-1) + (dx + L-1); the momentum-space occupation is
 * their Fourier sum (finishMeasurements does it on the host). */
int dqmc_measure_reset(dqmc_ctx* ctx);
int dqmc_measure_slice(dqmc_ctx* ctx);
size_t dqmc_measure_accum_size(dqmc_ctx* ctx);
int dqmc_measure_read_host(dqmc_ctx* ctx, double* out);
/* set_exchange_parameter_value (detsdwopdim.cpp:5195-5197): r only enters the bosonic action */
int dqmc_set_exchange_parameter(dqmc_ctx* ctx, double r);

/* ---- measurement helpers for bench.py -------------------------------------------------------- */
/* Device time per kernel family, measured with HIP events on the context's own stream while profiling is
 * switched on, plus launch counts and decomposition statistics. */
enum { DQMC_FAM_BMULT = 0, DQMC_FAM_GEMM = 1, DQMC_FAM_DECOMP = 2, DQMC_FAM_DECIDE = 3, DQMC_FAM_OTHER = 4,
       DQMC_FAM_GATHER = 5, DQMC_FAM_FLUSH = 6, DQMC_FAM_COUNT = 8 };
typedef struct dqmc_profile {
    double ms[8];              /* per family: bmult, gemm (fixed-size products), decomp (Jacobi rounds or QR),
                                  decide, other, gather, flush (G += X Gr), unused */
    uint64_t launches[8];
    uint64_t svd_calls, svd_sweeps_total, svd_sweeps_max, qr_calls;
    double gemm_flops;         /* 8 M N K summed over the launches of family gemm (4 M N K for the triangular chaining product) */
    double decomp_round_ms;    /* SVD mode: time inside batches of back-to-back Jacobi rounds only */
    uint64_t decomp_rounds;
    uint64_t blocks_nonempty;   /* delayed-update blocks that really flushed, summed over all chains, since dqmc_profile_enable */
    uint64_t chains;            /* chains every launch of this context carries */
    uint64_t updates_accepted;  /* accepted local updates, summed over all chains, since dqmc_profile_enable (flush flops = 8 n_g^2 MSF each) */
    uint64_t lu_calls;          /* QR mode: Green's functions whose inner inverse came from the LU factorisation (n_g <= 512); qr_calls then
                                   counts the chain factorisations (UDT) only */
    /* GPU-filling launches INSIDE the decomposition family, timed on their own (their time is also part of ms[DQMC_FAM_DECOMP]):
       [0] trailing updates of the LU factorisation (K = 32, on the flush kernel), [1] products inside factorisations and triangular
       solves (k_zgemm: the levels of the recursive solves, the block Gram-Schmidt QR); flops counted as 8 M N K, bytes as the
       operands read once and the result written (read-modify-written) once */
    double sub_ms[4];
    uint64_t sub_launches[4];
    double sub_flops[4];
    double sub_bytes[4];
} dqmc_profile;
int dqmc_profile_enable(dqmc_ctx* ctx, int on);
int dqmc_profile_read(dqmc_ctx* ctx, dqmc_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* DQMC_HIP_H_ */
