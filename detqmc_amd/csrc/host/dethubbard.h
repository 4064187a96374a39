// Host-side Hubbard replica: the build's DetHubbard (reference src/dethubbard.{h,cpp}) -- pure control flow over the
// kernel ABI, like detsdw.h.  nb >= 1 independent replicas advance in lockstep through one kernel context.
#pragma once
#include <string>
#include <vector>
#include "../../../include/dethubbard_host.h"
#include "detsdw.h"             // GeneralError / ParameterWrong, RngStream

namespace detqmc {

class DetHubbard {
public:
    DetHubbard(const dethubbard_params& pars, int nchains);
    ~DetHubbard();
    DetHubbard(const DetHubbard&) = delete;
    DetHubbard& operator=(const DetHubbard&) = delete;

    int numChains() const { return (int)ch_.size(); }
    void sweep(bool takeMeasurements);
    void sweepThermalization() { sweep_skeleton(false); }
    void getInfo(dethubbard_info& out, int b);
    void getAuxfield(double* out, int b);
    void getGreen(double* gUp, double* gDn, int b);
    void getObservables(dethubbard_observables& out, int b) const { out = ch_[b].obs; }
    void getZcorr(double* out, int b) const;
    void saveState(const std::string& path);
    void loadState(const std::string& path);
    double rand01(int b) { return ch_[b].rng.rand01(); }
    dqmc_ctx* ctx() { return ctx_; }

private:
    enum SweepDirection { Up = +1, Down = -1 };
    struct Chain {
        RngStream rng;
        std::vector<double> aux;               // host mirror of the auxiliary field [m+1][N], +-1.0
        double lastAccRatio = 0.0;
        dethubbard_observables obs{};
        std::vector<double> zcorr;
        Chain(uint32_t seed, uint32_t simindex) : rng(seed, simindex + 1u) {}      // detqmc.h:181
    };
    dethubbard_params p_;
    std::vector<Chain> ch_;
    std::vector<double> window_;
    int N_, m_, s_, n_;
    double alpha_;
    dqmc_ctx* ctx_ = nullptr;
    SweepDirection lastSweepDir_ = Up;
    int performedSweeps_ = 0;
    bool measuring_ = false;

    static void check(int rc, const char* what);
    void sweep_skeleton(bool takeMeasurements);
    void sweepDown();
    void sweepUp();
    void updateInSlice(int k);
    void finishMeasurements(int b);
};

}  // namespace detqmc
