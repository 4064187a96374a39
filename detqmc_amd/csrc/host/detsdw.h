// Host-side replica: the build's equivalent of DetSDW<CB_ASSAAD_BERG, OPDIM> on top of
// DetModelGC<1, cpx> (reference src/detsdwopdim.{h,cpp}, src/detmodel.h).  Pure control flow: every
// numerical step is a call into the kernel ABI (include/dqmc_hip.h).  Method names follow the
// reference so the parity tests read like the reference's own call sites.
#pragma once
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../../include/detsdw_host.h"
#include "dsfmt19937.h"

namespace detqmc {

// src/exceptions.h:24-73
struct GeneralError : std::runtime_error {
    int code;
    GeneralError(int code_, const std::string& msg) : std::runtime_error(msg), code(code_) {}
};
struct ParameterWrong : GeneralError {
    explicit ParameterWrong(const std::string& msg) : GeneralError(DQMC_EINVAL, msg) {}
};

// One object drives nb >= 1 replicas ("chains") that share lattice, temperature and couplings and differ in
// RNG stream (rngSeed / simindex), field configuration and the exchange parameter r -- exactly the set of
// replicas of a DetQMCPT run (src/detqmcpt.h).  nb = 1 is the reference's single replica.
// Per-replica methods take the chain index b.
//
// The chains are held in S sub-batches ("groups"), each ONE kernel context (dqmc_create_batch: every launch carries
// all chains of the group, own HIP stream).  A sweep runs the groups concurrently, one host thread per group: the
// groups drift out of phase, so the latency-bound kernels of one (decision kernel, QR panel: one workgroup per chain)
// overlap the streaming / MFMA kernels of the others -- within one process what round 1 needed four worker
// processes for.  Chains are independent Markov chains, so the grouping changes no result (tests).
class DetSDW {
public:
    explicit DetSDW(const detsdw_params& pars) : DetSDW(&pars, 1) {}     // createReplica + ctor
    // sub_batches: 0 = automatic (up to 4 groups of at least 32 chains each), otherwise a divisor of nchains
    DetSDW(const detsdw_params* pars, int nchains, int sub_batches = 0);
    ~DetSDW();
    DetSDW(const DetSDW&) = delete;
    DetSDW& operator=(const DetSDW&) = delete;

    int numChains() const { return (int)ch_.size(); }
    void sweep(bool takeMeasurements);
    void sweepThermalization();

    // replica exchange surface
    double get_exchange_parameter_value(int b = 0) const { return ch_[b].pars.r; }
    void set_exchange_parameter_value(double r, int b = 0);
    const char* get_exchange_parameter_name() const { return "r"; }
    double get_exchange_action_contribution(int b = 0);
    void get_control_data(detsdw_control_data& out, int b = 0);
    void set_control_data(const detsdw_control_data& in, int b = 0);

    void getInfo(detsdw_info& out, int b = 0);
    void getObservables(detsdw_observables& out, int b = 0) const { out = ch_[b].obs; }
    void getObservableVector(int which, double* out, int b = 0) const;
    void getPhi(double* phi, int b = 0);
    void setPhi(const double* phi, int b = 0);
    void getCdwl(int32_t* cdwl, int b = 0);
    void setCdwl(const int32_t* cdwl, int b = 0);
    void getGreen(dqmc_cplx* g, int b = 0);
    void getGreenInvSv(double* sv, int b = 0);
    void saveConfigurationStreamBinary(const std::string& directory, int b = 0);
    void saveState(const std::string& path);
    void loadState(const std::string& path);
    double rand01(int b = 0) { return ch_[b].rng.rand01(); }
    int numSubBatches() const { return (int)groups_.size(); }
    // kernel context that holds chain b and b's index inside it
    dqmc_ctx* ctx(int b = 0, int* local = nullptr) { if (local) *local = b - grp(b).first; return grp(b).ctx; }

private:
    enum SweepDirection { Up = +1, Down = -1 };
    struct Chain {
        detsdw_params pars;
        RngStream rng;
        std::vector<double> phi;               // host mirror, valid after syncPhiFromDevice(b)
        std::vector<int32_t> cdwl;             // discrete field l_i(tau_k), [k * N + site]; lives on the device when cdwU != 0 (getCdwl)
        int acceptedGlobalShifts = 0, attemptedGlobalShifts = 0;          // UpdateStatistics (detsdwopdim.h:285-311)
        int acceptedWolffClusterUpdates = 0, attemptedWolffClusterUpdates = 0;
        int acceptedWolffClusterShiftUpdates = 0, attemptedWolffClusterShiftUpdates = 0;
        double addedWolffClusterSize = 0.0;
        double phiDelta = 0.5, lastAccRatio = 0.0;
        double angleDelta = 0.0, scaleDelta = 0.1;          // AdjustmentData::InitialAngleDelta / InitialScaleDelta (detsdwopdim.h:490-491)
        detsdw_observables obs{};
        std::vector<double> kOccX, kOccY, pairPlus, pairMinus;
        Chain(const detsdw_params& p) : pars(p), rng(p.rngSeed, (uint32_t)p.simindex + 1u) {}   // detqmc.h:181
    };
    std::vector<Chain> ch_;
    struct Group {                                 // one sub-batch = one kernel context, chains [first, first + count)
        dqmc_ctx* ctx = nullptr;
        int first = 0, count = 0;
        std::vector<double> window;                // uniforms of the coming sweep, all chains of the group
        std::vector<double> fields;                // host staging of all chains' fields (global moves)
    };
    std::vector<Group> groups_;
    int N_, MSF_, ng_, m_, s_, n_, opdim_;
    SweepDirection lastSweepDir_ = Up;
    int performedSweeps_ = 0;

    Group& grp(int b) { return groups_[(size_t)b / (size_t)groups_[0].count]; }
    static void check(int rc, const char* what);
    dqmc_ctx* select(int b);                       // selects chain b in its context and returns that context
    static void normalise(detsdw_params& p, int& bcv);
    void setupRandomField(Chain& c);
    void setupUdVStorage_and_calculateGreen(Group& g);
    void forEachGroup(const std::function<void(Group&)>& fn);      // concurrently, one host thread per group
    void sweep_skeleton(Group& g, bool thermalization);
    void measureBosonic(Chain& c, bool descending);
    void finishFermionic(int b);
    bool measuring_ = false;          // measure(k) after the updates of slice k (updateInSliceAndMaybeMeasure)
    void sweepDown(Group& g, bool thermalization);
    void sweepUp(Group& g, bool thermalization);
    void updateInSlice(Group& g, int k, bool thermalization);
    int uniformsPerSite() const;
public:
    void exchangeActionsDevice(double* out_dev);        // out_dev[chain], device memory
private:
    void beginLocalUpdates(Group& g);
    void endLocalUpdates(Group& g);
    void globalMove(Group& g);
    enum GlobalMoveKind { MoveShift, MoveWolff, MoveWolffShift };
    void attemptGlobalMove(Group& g, GlobalMoveKind kind);
    unsigned buildAndFlipCluster(Chain& c);
    void addGlobalRandomDisplacement(Chain& c);
    double phiAction(const Chain& c) const;
    void syncPhiFromDevice(int b);
    size_t phiIdx(int site, int dim, int k) const { return (size_t)site + (size_t)N_ * (dim + (size_t)opdim_ * k); }
};

}  // namespace detqmc
