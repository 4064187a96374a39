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

class DetSDW {
public:
    explicit DetSDW(const detsdw_params& pars);     // createReplica + ctor
    ~DetSDW();
    DetSDW(const DetSDW&) = delete;
    DetSDW& operator=(const DetSDW&) = delete;

    void sweep(bool takeMeasurements);
    void sweepThermalization();

    // replica exchange surface
    double get_exchange_parameter_value() const { return pars_.r; }
    void set_exchange_parameter_value(double r);
    const char* get_exchange_parameter_name() const { return "r"; }
    double get_exchange_action_contribution();
    void get_control_data(detsdw_control_data& out);
    void set_control_data(const detsdw_control_data& in);

    void getInfo(detsdw_info& out);
    void getPhi(double* phi);
    void setPhi(const double* phi);
    void getGreen(dqmc_cplx* g);
    void getGreenInvSv(double* sv);
    double rand01() { return rng_.rand01(); }
    dqmc_ctx* ctx() { return ctx_; }

private:
    enum SweepDirection { Up = +1, Down = -1 };
    detsdw_params pars_;
    int N_, MSF_, ng_, m_, s_, n_, opdim_;
    RngStream rng_;
    dqmc_ctx* ctx_ = nullptr;
    SweepDirection lastSweepDir_ = Up;
    int performedSweeps_ = 0;
    int acceptedGlobalShifts_ = 0, attemptedGlobalShifts_ = 0;
    std::vector<double> phi_;                  // host mirror, valid after syncPhiFromDevice()
    double phiDelta_ = 0.5, lastAccRatio_ = 0.0;

    void check(int rc, const char* what);
    void setupRandomField();
    void setupUdVStorage_and_calculateGreen();
    void sweep_skeleton(bool thermalization);
    void sweepDown(bool thermalization);
    void sweepUp(bool thermalization);
    void updateInSlice(int k, bool thermalization);
    void beginLocalUpdates();
    void endLocalUpdates();
    void globalMove();
    void attemptGlobalShiftMove();
    double phiAction() const;
    void syncPhiFromDevice();
    double& phi(int site, int dim, int k) { return phi_[(size_t)site + (size_t)N_ * (dim + (size_t)opdim_ * k)]; }
    double phi(int site, int dim, int k) const { return phi_[(size_t)site + (size_t)N_ * (dim + (size_t)opdim_ * k)]; }
};

}  // namespace detqmc
