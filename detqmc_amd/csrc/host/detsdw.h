// Host-side replica: the build's equivalent of DetSDW<CB_ASSAAD_BERG, OPDIM> on top of
// DetModelGC<1, cpx> (reference src/detsdwopdim.{h,cpp}, src/detmodel.h).  Pure control flow: every
// numerical step is a call into the kernel ABI (include/dqmc_hip.h).  Method names follow the
// reference so the parity tests read like the reference's own call sites.
#pragma once
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
// replicas of a DetQMCPT run (src/detqmcpt.h).  All chains advance in lockstep through ONE kernel context
// (dqmc_create_batch), i.e. every launch carries all chains; nb = 1 is the reference's single replica.
// Per-replica methods take the chain index b.
class DetSDW {
public:
    explicit DetSDW(const detsdw_params& pars) : DetSDW(&pars, 1) {}     // createReplica + ctor
    DetSDW(const detsdw_params* pars, int nchains);
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
    void getGreen(dqmc_cplx* g, int b = 0);
    void getGreenInvSv(double* sv, int b = 0);
    void saveConfigurationStreamBinary(const std::string& directory, int b = 0);
    void saveState(const std::string& path);
    void loadState(const std::string& path);
    double rand01(int b = 0) { return ch_[b].rng.rand01(); }
    dqmc_ctx* ctx() { return ctx_; }

private:
    enum SweepDirection { Up = +1, Down = -1 };
    struct Chain {
        detsdw_params pars;
        RngStream rng;
        std::vector<double> phi;               // host mirror, valid after syncPhiFromDevice(b)
        int acceptedGlobalShifts = 0, attemptedGlobalShifts = 0;          // UpdateStatistics (detsdwopdim.h:285-311)
        int acceptedWolffClusterUpdates = 0, attemptedWolffClusterUpdates = 0;
        int acceptedWolffClusterShiftUpdates = 0, attemptedWolffClusterShiftUpdates = 0;
        double addedWolffClusterSize = 0.0;
        double phiDelta = 0.5, lastAccRatio = 0.0;
        detsdw_observables obs{};
        std::vector<double> kOccX, kOccY, pairPlus, pairMinus;
        Chain(const detsdw_params& p) : pars(p), rng(p.rngSeed, (uint32_t)p.simindex + 1u) {}   // detqmc.h:181
    };
    std::vector<Chain> ch_;
    int N_, MSF_, ng_, m_, s_, n_, opdim_;
    dqmc_ctx* ctx_ = nullptr;
    SweepDirection lastSweepDir_ = Up;
    int performedSweeps_ = 0;

    void check(int rc, const char* what);
    void select(int b);
    static void normalise(detsdw_params& p, int& bcv);
    void setupRandomField(Chain& c);
    void setupUdVStorage_and_calculateGreen();
    void sweep_skeleton(bool thermalization);
    void measureBosonic(Chain& c, bool descending);
    void finishFermionic(int b);
    bool measuring_ = false;          // measure(k) after the updates of slice k (updateInSliceAndMaybeMeasure)
    void sweepDown(bool thermalization);
    void sweepUp(bool thermalization);
    void updateInSlice(int k, bool thermalization);
    void beginLocalUpdates();
    void endLocalUpdates();
    void globalMove();
    enum GlobalMoveKind { MoveShift, MoveWolff, MoveWolffShift };
    void attemptGlobalMove(GlobalMoveKind kind);
    unsigned buildAndFlipCluster(Chain& c);
    void addGlobalRandomDisplacement(Chain& c);
    double phiAction(const Chain& c) const;
    void syncPhiFromDevice(int b);
    size_t phiIdx(int site, int dim, int k) const { return (size_t)site + (size_t)N_ * (dim + (size_t)opdim_ * k); }
};

}  // namespace detqmc
