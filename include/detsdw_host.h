/*
 * detsdw_host.h -- C API of the C++ host layer that sits ABOVE the kernel ABI (dqmc_hip.h).
 *
 * The host layer (detqmc_amd/csrc/host/detsdw.{h,cpp}) is the build's DetSDW / DetModelGC
 * equivalent: it owns the RNG stream, the parameter checks and the control flow of
 * sweep_skeleton / sweepUp / sweepDown / globalMove (reference src/detmodel.h:1266-1478,
 * src/detsdwopdim.cpp:3461-3644, 4423-4502) and calls ONLY the C ABI of dqmc_hip.h.  This header
 * exposes that C++ class with the reference's method names so that harnesses written in C, or
 * Python through ctypes, can drive it; it mirrors the operator surface DetQMC<Model> /
 * DetQMCPT<Model> require of a replica (src/detqmc.h:58-156, src/detmodel.h:138-156,
 * src/detsdwopdim.h:116-153).
 */
#ifndef DETSDW_HOST_H_
#define DETSDW_HOST_H_

#include <stddef.h>
#include <stdint.h>
#include "dqmc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct detsdw_replica detsdw_replica;

/* ModelParamsDetSDW (src/detsdwparams.h:24-120) + rngSeed/simindex of DetQMCParams
 * (src/detqmcparams.h) as far as the sweep path uses them.  Unsupported reference options
 * (turnoffFermions, overRelaxation, phiFixed) are rejected by detsdw_create with DQMC_EINVAL and a
 * message naming the option. */
typedef struct detsdw_params {
    int32_t opdim;
    int32_t L;
    int32_t m;                   /* give m > 0 OR beta > 0, not both (detmodelparams.h:97-110) */
    int32_t s;
    int32_t delaySteps;
    int32_t globalShift;
    int32_t globalUpdateInterval;
    int32_t weakZflux;
    int32_t phi2bosons;
    int32_t device;
    int32_t simindex;
    uint32_t rngSeed;
    int32_t has_mux_muy;         /* if 0: mux = muy = mu (detsdwopdim.cpp:75-79) */
    int32_t updateMethod;        /* 0 iterative, 1 woodbury, 2 delayed (all give the same chain; 0/1 run with D = 1) */
    char bc[16];                 /* "pbc", "apbc-x", "apbc-y", "apbc-xy" */
    double beta, dtau;
    double r, c, u, lambda;
    double txhor, txver, tyhor, tyver;
    double mu, mux, muy;
    double accRatio;
    double cdwU;                 /* != 0: discrete field l_i(tau) next to phi, updated in a second pass over each slice (detsdwopdim.cpp:2474-2485) */
    int32_t stabilisation;       /* 0 = SVD (as the reference), 1 = QR/UDT (same G to rounding, much faster) */
    int32_t cb_none;             /* 0 = checkerboard (default), 1 = checkerboard=false: dense B matrices (CB_NONE) */
    int32_t wolffClusterUpdate;       /* attemptWolffClusterUpdate every globalUpdateInterval sweeps (detsdwopdim.cpp:3488-3562) */
    int32_t wolffClusterShiftUpdate;  /* combined cluster + global shift (:3647-3751); excludes the two individual moves */
    int32_t repeatWolffPerSweep;      /* cluster flips per attempt, 0 is read as 1 */
    int32_t fermionMeasurements;      /* 1: sweep(takeMeasurements) also takes the G-dependent observables (the reference's
                                         default, i.e. turnoffFermionMeasurements = false); 0: bosonic observables only */
    int32_t spinProposalMethod;       /* 0 box (default), 1 rotate_then_scale, 2 rotate_and_scale -- the latter two for opdim == 3 only
                                         (src/detsdwparams.h:40-42, src/detsdwopdim.cpp:2447-2470) */
    int32_t adaptScaleVariance;       /* adapt scaleDelta during thermalization (src/detsdwparams.h:43) */
    int32_t repeatUpdateInSlice;      /* passes of local updates per time slice and sweep, 0 is read as 1 (src/detsdwparams.h:90) */
    int32_t reserved_model;
    dqmc_tuning tuning;               /* result-neutral execution choices handed to every kernel context (dqmc_hip.h); all zero =
                                         automatic.  With pipeline = 0 the host layer switches the pipelined update on only for
                                         handles of at most two kernel contexts (more contexts overlap each other instead) */
} detsdw_params;

typedef struct detsdw_info {
    int32_t opdim, L, N, MSF, n_g, m, s, n;
    int32_t performedSweeps;
    int32_t lastSweepDir;        /* +1 up, -1 down */
    int32_t acceptedGlobalShifts, attemptedGlobalShifts;
    int32_t currentTimeslice;
    int32_t reserved;
    int32_t acceptedWolffClusterUpdates, attemptedWolffClusterUpdates;
    int32_t acceptedWolffClusterShiftUpdates, attemptedWolffClusterShiftUpdates;
    double addedWolffClusterSize;
    double beta, dtau;
    double phiDelta, lastAccRatioLocal_phi;
    double r;                    /* exchange parameter */
    double angleDelta, scaleDelta;   /* rotate / scale proposals (AdjustmentData, src/detsdwopdim.h:510-512) */
    uint64_t rngDrawn;           /* uniforms consumed from the stream so far */
} detsdw_info;

/* control data swapped in replica exchange: UpdateStatistics + AdjustmentData
 * (src/detsdwopdim.cpp:5219-5247), fixed-size POD instead of a boost archive */
typedef struct detsdw_control_data {
    int32_t acceptedGlobalShifts, attemptedGlobalShifts;
    int32_t acceptedWolffClusterUpdates, attemptedWolffClusterUpdates;
    int32_t acceptedWolffClusterShiftUpdates, attemptedWolffClusterShiftUpdates;
    double addedWolffClusterSize;
    dqmc_update_state adjust;
} detsdw_control_data;

/* bosonic observables of a measurement sweep: initMeasurements / measure / finishMeasurements with
 * turnoffFermionMeasurements (src/detsdwopdim.cpp:441-456, :509-545, :903-921); valid after detsdw_sweep(r, 1).
 * The fermionic observables (pairing, k-space occupation, ...) are not built yet. */
typedef struct detsdw_observables {
    double meanPhi[3];
    double normMeanPhi;
    double associatedEnergy;
    double phiRhoS_Gc, phiRhoS_Gs;      /* opdim == 2 only */
    int32_t valid;                      /* 1 after a sweep with takeMeasurements */
    int32_t fermionic_valid;            /* 1 if the fields below and the vectors of detsdw_get_observable_vector were taken */
    double greenK0, greenLocal;         /* src/detsdwopdim.cpp:565-588, :926-927 */
    double pairPlusMax, pairMinusMax;   /* :986-1002 */
    double occDiffSq;                   /* :866-897, :1013 */
} detsdw_observables;
enum { DETSDW_OBS_KOCCX = 0, DETSDW_OBS_KOCCY = 1, DETSDW_OBS_PAIRPLUS = 2, DETSDW_OBS_PAIRMINUS = 3 };

/* createReplica (src/detsdwopdim.cpp:49-84) + DetSDW ctor (:158-361): checks parameters, seeds the
 * RNG with (rngSeed, simindex + 1) (src/detqmc.h:181), draws the random field, builds UdV storage and
 * G(beta) */
int detsdw_create(const detsdw_params* p, detsdw_replica** out);
/* The replicas of one parallel-tempering ensemble held by ONE process / GPU (the reference holds one replica per
 * MPI rank, src/detqmcpt.h:300-420): p[0..nchains) may differ only in r, rngSeed and simindex.  All chains
 * sweep in lockstep (every kernel launch carries all of them); each follows exactly the Markov chain a
 * single replica created from p[b] would.  detsdw_sweep* act on all chains, every other call below on the
 * chain chosen with detsdw_select_chain (default 0). */
int detsdw_create_batch(const detsdw_params* p, int nchains, detsdw_replica** out);
/* The chains of a batch live in `sub_batches` kernel contexts (own HIP stream each) that a sweep drives concurrently, one
 * host thread per context: the contexts drift out of phase, so the latency-bound kernels of one overlap the streaming /
 * MFMA kernels of the others inside ONE process.  Results do not depend on the grouping (independent Markov chains).
 * sub_batches = 0 (what detsdw_create_batch passes): automatic, up to 4 contexts of at least 32 chains each; otherwise a
 * divisor of nchains. */
int detsdw_create_batch_ex(const detsdw_params* p, int nchains, int sub_batches, detsdw_replica** out);
int detsdw_num_sub_batches(detsdw_replica* r);
int detsdw_select_chain(detsdw_replica* r, int chain);
int detsdw_num_chains(detsdw_replica* r);
void detsdw_destroy(detsdw_replica* r);
const char* detsdw_last_error(void);

/* DetModel::sweep / sweepThermalization (src/detmodel.h:138-147, src/detsdwopdim.cpp:4423-4502) */
int detsdw_sweep(detsdw_replica* r, int takeMeasurements);
int detsdw_sweep_thermalization(detsdw_replica* r);

int detsdw_get_info(detsdw_replica* r, detsdw_info* out);
int detsdw_get_observables(detsdw_replica* r, detsdw_observables* out);
/* N-vectors of the last measurement sweep: kOccX, kOccY (site index = k-vector, :616-659, :937-941), pairPlus, pairMinus */
int detsdw_get_observable_vector(detsdw_replica* r, int which, double* out);
/* phi in the reference layout (N, OPDIM, m+1) column-major */
int detsdw_get_phi(detsdw_replica* r, double* phi);
int detsdw_set_phi(detsdw_replica* r, const double* phi);      /* also rebuilds UdV storage and G */
/* the discrete field cdwl(site, k) in the reference layout (N x (m+1) column-major, values +-1 / +-2, slice 0 unused);
 * set needs cdwU != 0 and rebuilds UdV storage and G */
int detsdw_get_cdwl(detsdw_replica* r, int32_t* cdwl);
int detsdw_set_cdwl(detsdw_replica* r, const int32_t* cdwl);
int detsdw_get_green(detsdw_replica* r, dqmc_cplx* g);
int detsdw_get_green_inv_sv(detsdw_replica* r, double* sv);
double detsdw_rng_rand01(detsdw_replica* r);                   /* draws from the replica's stream */
dqmc_ctx* detsdw_ctx(detsdw_replica* r);                       /* kernel context holding the selected chain */
dqmc_ctx* detsdw_ctx_of_chain(detsdw_replica* r, int chain, int* local_index);   /* ... and the chain's index inside it */

/* saveConfigurationStreamBinary (src/detsdwopdim.cpp:4991-5012): appends the current field configuration to
 * <directory>/configs-phi.binarystream in the reference's order (x outer, y, k = 1..m, component; raw fp64), the
 * format its evaluation tools (sdwcorr, deteval) read */
int detsdw_save_configuration_stream_binary(detsdw_replica* r, const char* directory);

/* Checkpoint / resume: field configurations, RNG stream positions, step-size adaptation state and update statistics of
 * every chain (what the reference keeps in simulation.state, src/detsdwopdim.h:1127-1148, src/rngwrapper.h:100-116; own
 * binary format).  detsdw_load_state needs a replica created with the same parameters; it rebuilds UdV storage and
 * G(beta) like the reference's resume does, i.e. the next sweep is a down sweep -- a checkpoint written after an even
 * number of sweeps continues exactly the uninterrupted Markov chain. */
int detsdw_save_state(detsdw_replica* r, const char* path);
int detsdw_load_state(detsdw_replica* r, const char* path);

/* replica-exchange surface (src/detsdwopdim.h:116-153, src/detsdwopdim.cpp:5185-5247) */
double detsdw_get_exchange_parameter_value(detsdw_replica* r);
int detsdw_set_exchange_parameter_value(detsdw_replica* r, double value);
const char* detsdw_get_exchange_parameter_name(detsdw_replica* r);
int detsdw_get_exchange_action_contribution(detsdw_replica* r, double* out);
/* all chains of the handle at once, on the device: out_dev[chain] (device array of detsdw_num_chains doubles) */
int detsdw_exchange_actions_device(detsdw_replica* r, double* out_dev);
int detsdw_get_control_data(detsdw_replica* r, detsdw_control_data* out);
int detsdw_set_control_data(detsdw_replica* r, const detsdw_control_data* in);
/* get_replica_exchange_probability<DetSDW> (src/detsdwopdim.cpp:5251-5264) */
double detsdw_replica_exchange_probability(double par1, double action1, double par2, double action2);

/* RNG restatement, exposed for host-only tests: first n draws of RngWrapper(seed, processIndex) */
int detsdw_rng_fill(uint32_t seed, uint32_t processIndex, double* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* DETSDW_HOST_H_ */
