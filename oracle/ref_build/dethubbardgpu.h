// Reference-side binding of libdetqmc_amd.so for the Hubbard replica (BASELINE config 1): the model class that lets the
// reference's driver template DetQMC<Model, ModelParams> (src/detqmc.h) run its DetHubbard on an MI355X.
// BUILT ONLY WHERE /root/reference EXISTS (Makefile target `detqmchubbardgpu`); test infrastructure, copies nothing.
// Same cut as detsdwgpu.h; the surface is the one DetHubbard offers (src/dethubbard.h:58-100, src/dethubbard.cpp:37-118):
// createReplica, the eight scalar observables and the spin-z correlation vector in the reference's names and order,
// sweep / sweepThermalization, prepareModelMetadataMap, saveContents / loadContents; configuration streams are "not
// implemented" in the reference and here.
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>
#include <armadillo>
#include "boost/serialization/string.hpp"
#include "boost/serialization/split_member.hpp"
#include "detmodelloggingparams.h"
#include "detmodelparams.h"
#include "dethubbardparams.h"
#include "exceptions.h"
#include "metadata.h"
#include "observable.h"
#include "rngwrapper.h"
#include "tools.h"
#include "dethubbard_host.h"        // -I <repo>/include

class DetHubbardGpu {
public:
    typedef ModelParams<DetHubbard> ModelParamsT;
    friend void createReplica(std::unique_ptr<DetHubbardGpu>& replica_out, RngWrapper& rng, ModelParams<DetHubbard> pars,
                              DetModelLoggingParams loggingPars);
    ~DetHubbardGpu() { dethubbard_destroy(h_); }
    DetHubbardGpu(const DetHubbardGpu&) = delete;
    DetHubbardGpu& operator=(const DetHubbardGpu&) = delete;

    uint32_t getSystemN() const { return N_; }
    // src/dethubbard.cpp:124-146
    MetadataMap prepareModelMetadataMap() const {
        MetadataMap meta = pars_.prepareMetadataMap();
        dethubbard_info i;
        call(dethubbard_get_info(h_, &i), "dethubbard_get_info");
        meta["N"] = numToString(N_);
        meta["alpha"] = numToString(i.alpha);
        return meta;
    }
    std::vector<ScalarObservable> getScalarObservables() { return obsScalar_; }
    std::vector<VectorObservable> getVectorObservables() { return obsVector_; }
    std::vector<KeyValueObservable> getKeyValueObservables() { return std::vector<KeyValueObservable>(); }

    void sweep(bool takeMeasurements) {
        const uint64_t before = drawn();
        call(dethubbard_sweep(h_, takeMeasurements ? 1 : 0), "sweep");
        advanceWrapper(drawn() - before);
        if (takeMeasurements) {
            dethubbard_observables o;
            call(dethubbard_get_observables(h_, &o), "dethubbard_get_observables");
            occUp = o.occUp; occDn = o.occDn; occTotal = o.occTotal; occDouble = o.occDouble; localMoment = o.localMoment;
            eKinetic = o.eKinetic; ePotential = o.ePotential; eTotal = o.eTotal;
            call(dethubbard_get_zcorr(h_, zcorr.memptr()), "dethubbard_get_zcorr");
        }
    }
    void sweepThermalization() {
        const uint64_t before = drawn();
        call(dethubbard_sweep_thermalization(h_), "sweepThermalization");
        advanceWrapper(drawn() - before);
    }
    void sweepSimple(bool) { throw_GeneralError("DetHubbardGpu: greenUpdate=simple is not supported, use greenUpdate=stabilized"); }
    void sweepSimpleThermalization() { sweepSimple(false); }
    void thermalizationOver() {}
    // src/dethubbard.h:74-93
    void saveConfigurationStreamText(const std::string& = ".") { throw_GeneralError("DetHubbard::saveConfigurationStreamText not implemented"); }
    void saveConfigurationStreamBinary(const std::string& = ".") { throw_GeneralError("DetHubbard::saveConfigurationStreamBinary not implemented"); }
    void saveConfigurationStreamTextHeader(const std::string&, const std::string& = ".") { throw_GeneralError("DetHubbard::saveConfigurationStreamTextHeader not implemented"); }
    void saveConfigurationStreamBinaryHeaderfile(const std::string&, const std::string& = ".") { throw_GeneralError("DetHubbard::saveConfigurationStreamBinaryHeaderfile not implemented"); }

    template<class Archive> void saveContents(Archive& ar) {
        TempFile tf;
        call(dethubbard_save_state(h_, tf.path.c_str()), "dethubbard_save_state");
        std::ifstream in(tf.path, std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        std::string blob = ss.str();
        ar & blob;
    }
    template<class Archive> void loadContents(Archive& ar) {
        std::string blob;
        ar & blob;
        TempFile tf;
        { std::ofstream out(tf.path, std::ios::binary); out.write(blob.data(), (std::streamsize)blob.size()); }
        call(dethubbard_load_state(h_, tf.path.c_str()), "dethubbard_load_state");
    }

private:
    struct TempFile {
        std::string path;
        TempFile() {
            char name[] = "/tmp/dethubbardgpu-state-XXXXXX";
            const int fd = mkstemp(name);
            if (fd < 0) throw_GeneralError(std::string("mkstemp: ") + strerror(errno));
            close(fd);
            path = name;
        }
        ~TempFile() { std::remove(path.c_str()); }
    };
    struct RngProbe {            // what RngWrapper::save hands to an archive (src/rngwrapper.h:100-105)
        std::vector<uint32_t> u;
        std::string state;
        RngProbe& operator<<(const uint32_t& v) { u.push_back(v); return *this; }
        RngProbe& operator<<(const std::string& s) { state = s; return *this; }
    };
    uint64_t drawn() const {
        dethubbard_info i;
        call(dethubbard_get_info(h_, &i), "dethubbard_get_info");
        return i.rngDrawn;
    }
    // the driver serialises ITS wrapper (DetQMC::serializeContentsCommon): keep it at the replica's stream position
    void advanceWrapper(uint64_t n) { for (uint64_t i = 0; i < n; ++i) (void)rng_->rand01(); }

    DetHubbardGpu(RngWrapper& rng, const ModelParams<DetHubbard>& pars) : pars_(pars), rng_(&rng) {
        RngProbe probe;
        boost::serialization::detail::member_saver<RngProbe, const RngWrapper>::invoke(probe, rng, 0);
        if (probe.u.size() != 2 || probe.u[1] == 0) throw_GeneralError("DetHubbardGpu: unexpected RngWrapper state");
        {
            RngWrapper fresh(probe.u[0], probe.u[1]);
            RngProbe p2;
            boost::serialization::detail::member_saver<RngProbe, const RngWrapper>::invoke(p2, fresh, 0);
            if (p2.state != probe.state) throw_GeneralError("DetHubbardGpu: createReplica needs a freshly seeded RngWrapper");
        }
        dethubbard_params p;
        std::memset(&p, 0, sizeof(p));
        p.L = (int32_t)pars.L; p.d = (int32_t)pars.d; p.m = (int32_t)pars.m; p.s = (int32_t)pars.s;     // after updateTemperatureParameters
        p.checkerboard = pars.checkerboard ? 1 : 0;
        p.device = std::getenv("DQMC_DEVICE") ? std::atoi(std::getenv("DQMC_DEVICE")) : 0;
        p.simindex = (int32_t)probe.u[1] - 1; p.rngSeed = probe.u[0];
        const char* stab = std::getenv("DQMC_STABILISATION");
        p.stabilisation = (stab && std::string(stab) == "qr") ? 1 : 0;          // default: SVD like the reference
        p.beta = 0.0; p.dtau = pars.dtau; p.t = pars.t; p.U = pars.U; p.mu = pars.mu;
        if (dethubbard_create(&p, 1, &h_) != DQMC_OK) throw_GeneralError(std::string("dethubbard_create: ") + dethubbard_last_error());
        N_ = (uint32_t)std::pow((double)pars.L, (double)pars.d);
        advanceWrapper(drawn());                                 // the draws of setupRandomAuxfield
        using std::cref;      // src/dethubbard.cpp:85-96: names and order
        obsScalar_.push_back(ScalarObservable(cref(occUp), "occupationUp", "nUp"));
        obsScalar_.push_back(ScalarObservable(cref(occDn), "occupationDown", "nDown"));
        obsScalar_.push_back(ScalarObservable(cref(occTotal), "totalOccupation", "n"));
        obsScalar_.push_back(ScalarObservable(cref(occDouble), "doubleOccupation", "n2"));
        obsScalar_.push_back(ScalarObservable(cref(localMoment), "localMoment", "m^2"));
        obsScalar_.push_back(ScalarObservable(cref(eKinetic), "kineticEnergy", "e_t"));
        obsScalar_.push_back(ScalarObservable(cref(ePotential), "potentialEnergy", "e_U"));
        obsScalar_.push_back(ScalarObservable(cref(eTotal), "totalEnergy", "e"));
        zcorr.zeros(N_);
        obsVector_.push_back(VectorObservable(cref(zcorr), N_, "spinzCorrelationFunction", "zcorr"));
    }
    void call(int rc, const char* what) const {
        if (rc != DQMC_OK) throw_GeneralError(std::string(what) + ": " + dethubbard_last_error());
    }

    ModelParams<DetHubbard> pars_;
    RngWrapper* rng_;
    dethubbard_replica* h_ = nullptr;
    uint32_t N_ = 0;
    std::vector<ScalarObservable> obsScalar_;
    std::vector<VectorObservable> obsVector_;
    num occUp = 0, occDn = 0, occTotal = 0, occDouble = 0, localMoment = 0, eKinetic = 0, ePotential = 0, eTotal = 0;
    arma::Col<num> zcorr;
};

// src/dethubbard.cpp:37-47
inline void createReplica(std::unique_ptr<DetHubbardGpu>& replica_out, RngWrapper& rng, ModelParams<DetHubbard> pars,
                          DetModelLoggingParams /*ignored*/ = DetModelLoggingParams()) {
    pars = updateTemperatureParameters(pars);
    pars.check();
    replica_out = std::unique_ptr<DetHubbardGpu>(new DetHubbardGpu(rng, pars));
}
