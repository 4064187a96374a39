// Reference-side binding of libdetqmc_amd.so: the model class a maintainer of crstnbr/detqmc adds so that the
// reference's own driver template DetQMC<Model, ModelParams> (src/detqmc.h:58-156) runs its replica on an MI355X.
//
// BUILT ONLY WHERE /root/reference EXISTS (this header includes the reference's headers; it lives under oracle/ref_build
// with the other code that compiles against the reference tree, see Makefile target `detqmcsdwgpu`).  It is not part of
// the product and copies nothing from the reference: DetSDWGpu has the members DetQMC<> dereferences --
//   createReplica(...)                                         src/detqmc.h:183     (src/detsdwopdim.cpp:49-84)
//   get{Scalar,Vector,KeyValue}Observables()                   src/detqmc.h:190-207 (src/detmodel.h:127-133, src/observable.h:29-42)
//   sweep / sweepThermalization / sweepSimple* / thermalizationOver   src/detqmc.h:455-486 (src/detmodel.h:138-156)
//   prepareModelMetadataMap()                                  src/detqmc.h:338     (src/detsdwopdim.cpp:395-440)
//   saveContents / loadContents(Archive&)                      src/detqmc.h:128-135 (src/detsdwopdim.h:1127-1148)
//   saveConfigurationStream{Text,Binary}[Header[file]]         src/detqmc.h:212-218, 491-497 (src/detsdwopdim.cpp:4943-5110)
// -- and forwards every one of them to the C ABI of include/detsdw_host.h.  All numerics run on the GPU.
#pragma once
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>
#include <armadillo>
#include "boost/filesystem.hpp"
#include "boost/serialization/string.hpp"
#include "boost/serialization/split_member.hpp"
#include "detmodelloggingparams.h"
#include "detmodelparams.h"
#include "detsdwparams.h"
#include "exceptions.h"
#include "metadata.h"
#include "observable.h"
#include "rngwrapper.h"
#include "tools.h"
#include "detmodel.h"           // the primary template get_replica_exchange_probability<Model> (src/detmodel.h:97-108)
#include "detsdwsystemconfig.h"
#include "detsdwsystemconfigfilehandle.h"
#include "detsdw_host.h"        // -I <repo>/include

class DetSDWGpu {
public:
    typedef ModelParamsDetSDW ModelParams;

    friend void createReplica(std::unique_ptr<DetSDWGpu>& replica_out, RngWrapper& rng, ModelParamsDetSDW pars,
                              DetModelLoggingParams loggingPars, const std::string& logfiledir);

    ~DetSDWGpu() { detsdw_destroy(h_); }
    DetSDWGpu(const DetSDWGpu&) = delete;
    DetSDWGpu& operator=(const DetSDWGpu&) = delete;

    uint32_t getSystemN() const { return pars_.N; }

    // src/detsdwopdim.cpp:395-440: the parameters plus the acceptance ratios of the global moves
    MetadataMap prepareModelMetadataMap() const {
        MetadataMap meta = pars_.prepareMetadataMap();
        detsdw_info i;
        call(detsdw_get_info(h_, &i), "detsdw_get_info");
        auto ratio = [](double acc, double att) { return att > 0 ? acc / att : 0.0; };
        if (pars_.globalShift) meta["globalShiftAccRatio"] = numToString(ratio(i.acceptedGlobalShifts, i.attemptedGlobalShifts));
        if (pars_.wolffClusterUpdate) {
            meta["wolffClusterUpdateAccRatio"] = numToString(ratio(i.acceptedWolffClusterUpdates, i.attemptedWolffClusterUpdates));
            meta["averageAcceptedWolffClusterSize"] = numToString(ratio(i.addedWolffClusterSize, i.acceptedWolffClusterUpdates));
        }
        if (pars_.wolffClusterShiftUpdate) {
            meta["wolffClusterShiftUpdateAccRatio"] =
                numToString(ratio(i.acceptedWolffClusterShiftUpdates, i.attemptedWolffClusterShiftUpdates));
            meta["averageAcceptedWolffClusterSize"] = numToString(ratio(i.addedWolffClusterSize, i.acceptedWolffClusterShiftUpdates));
        }
        return meta;
    }

    std::vector<ScalarObservable> getScalarObservables() { return obsScalar_; }
    std::vector<VectorObservable> getVectorObservables() { return obsVector_; }
    std::vector<KeyValueObservable> getKeyValueObservables() { return obsKeyValue_; }

    void sweep(bool takeMeasurements) {
        rngBeforeSweep();
        call(detsdw_sweep(h_, takeMeasurements ? 1 : 0), "sweep");
        rngAfterSweep();
        if (takeMeasurements) refreshObservables();
    }
    void sweepThermalization() {
        rngBeforeSweep();
        call(detsdw_sweep_thermalization(h_), "sweepThermalization");
        rngAfterSweep();
    }
    // greenUpdate=simple recomputes G from scratch in every slice (src/detmodel.h:1480-1489): debugging aid, not accelerated
    void sweepSimple(bool) { throw_GeneralError("DetSDWGpu: greenUpdate=simple is not supported, use greenUpdate=stabilized"); }
    void sweepSimpleThermalization() { sweepSimple(false); }

    // src/detsdwopdim.cpp:4302-4362
    void thermalizationOver() { thermalizationOver(-1); }
    void thermalizationOver(int processIndex) {
        detsdw_info i;
        detsdw_control_data cd;
        call(detsdw_get_info(h_, &i), "detsdw_get_info");
        call(detsdw_get_control_data(h_, &cd), "detsdw_get_control_data");
        const std::string prefix = processIndex == -1 ? "" : "p" + numToString(processIndex) + ": r" + numToString(i.r) + " ";
        auto ratio = [](double acc, double att) { return att > 0 ? acc / att : 0.0; };
        std::cout << prefix << "After thermalization: phiDelta = " << i.phiDelta << '\n'
                  << prefix << "recent local accRatio = " << cd.adjust.ra_runningAverage << std::endl;
        if (pars_.globalShift)
            std::cout << prefix << "globalShiftMove acceptance ratio = " << ratio(i.acceptedGlobalShifts, i.attemptedGlobalShifts) << std::endl;
        if (pars_.wolffClusterUpdate)
            std::cout << prefix << "wolffClusterUpdate acceptance ratio = "
                      << ratio(i.acceptedWolffClusterUpdates, i.attemptedWolffClusterUpdates) << std::endl;
        if (pars_.wolffClusterShiftUpdate)
            std::cout << prefix << "wolffClusterShiftUpdate acceptance ratio = "
                      << ratio(i.acceptedWolffClusterShiftUpdates, i.attemptedWolffClusterShiftUpdates) << std::endl;
    }

    // ---- configuration streams (src/detsdwopdim.cpp:4943-5110) ----
    void saveConfigurationStreamBinary(const std::string& directory = ".") {
        call(detsdw_save_configuration_stream_binary(h_, directory.c_str()), "saveConfigurationStreamBinary");
    }
    void saveConfigurationStreamText(const std::string& directory = ".") {
        namespace fs = boost::filesystem;
        const fs::path p = fs::path(directory) / fs::path("configs-phi.textstream");
        std::ofstream out(p.c_str(), std::ios::app);
        if (!out) { std::cerr << "Could not open file " << p.string() << " for writing.\n"; return; }
        std::vector<double> phi((size_t)pars_.N * pars_.opdim * (pars_.m + 1));
        call(detsdw_get_phi(h_, phi.data()), "detsdw_get_phi");
        out.precision(14);
        out.setf(std::ios::scientific, std::ios::floatfield);
        for (uint32_t ix = 0; ix < pars_.L; ++ix)
            for (uint32_t iy = 0; iy < pars_.L; ++iy)
                for (uint32_t k = 1; k <= pars_.m; ++k)
                    for (uint32_t dim = 0; dim < pars_.opdim; ++dim)
                        out << phi[(iy * pars_.L + ix) + (size_t)pars_.N * (dim + (size_t)pars_.opdim * k)] << "\n";
        if (pars_.cdwU) {                   // configs-l.textstream (src/detsdwopdim.cpp:4967-4986)
            const fs::path pl = fs::path(directory) / fs::path("configs-l.textstream");
            std::ofstream lout(pl.c_str(), std::ios::app);
            if (!lout) { std::cerr << "Could not open file " << pl.string() << " for writing.\n"; return; }
            std::vector<int32_t> l((size_t)pars_.N * (pars_.m + 1));
            call(detsdw_get_cdwl(h_, l.data()), "detsdw_get_cdwl");
            for (uint32_t ix = 0; ix < pars_.L; ++ix)
                for (uint32_t iy = 0; iy < pars_.L; ++iy)
                    for (uint32_t k = 1; k <= pars_.m; ++k) lout << l[(iy * pars_.L + ix) + (size_t)pars_.N * k] << "\n";
        }
    }
    void saveConfigurationStreamTextHeader(const std::string& simInfoHeaderText, const std::string& directory = ".") {
        writeHeader(directory, "configs-phi.textstream", simInfoHeaderText, "## phi configuration stream\n");
        if (pars_.cdwU)                     // src/detsdwopdim.cpp:5055-5070
            writeHeader(directory, "configs-l.textstream", simInfoHeaderText, "## l configuration stream\n");
    }
    void saveConfigurationStreamBinaryHeaderfile(const std::string& simInfoHeaderText, const std::string& directory = ".") {
        writeHeader(directory, "configs-phi.infoheader", simInfoHeaderText,
                    "## binary phi configuration stream (64 bit double precision floats) in file configs-phi.binarystream\n");
        if (pars_.cdwU)                     // src/detsdwopdim.cpp:5092-5108
            writeHeader(directory, "configs-l.infoheader", simInfoHeaderText,
                        "## binary l configuration stream (32 bit signed integers) in file configs-l.binarystream\n");
    }

    // ---- replica-exchange surface (src/detsdwopdim.h:116-153), what DetQMCPT<Model> needs on top ----
    num get_exchange_parameter_value() const { return detsdw_get_exchange_parameter_value(h_); }
    void set_exchange_parameter_value(num r) { call(detsdw_set_exchange_parameter_value(h_, r), "set_exchange_parameter_value"); pars_.r = r; }
    const char* get_exchange_parameter_name() const { return detsdw_get_exchange_parameter_name(h_); }
    num get_exchange_action_contribution() const {
        double v = 0;
        call(detsdw_get_exchange_action_contribution(h_, &v), "get_exchange_action_contribution");
        return v;
    }
    void get_control_data(std::string& buffer) const {
        detsdw_control_data cd;
        call(detsdw_get_control_data(h_, &cd), "get_control_data");
        buffer.assign((const char*)&cd, sizeof(cd));
    }
    void set_control_data(const std::string& buffer) {
        if (buffer.size() != sizeof(detsdw_control_data)) throw_GeneralError("set_control_data: wrong buffer size");
        detsdw_control_data cd;
        std::memcpy(&cd, buffer.data(), sizeof(cd));
        call(detsdw_set_control_data(h_, &cd), "set_control_data");
    }

    // ---- what DetQMCPT<Model> needs to save system configurations per control parameter (src/detqmcpt.h:183-200, 665-760):
    //      the reference's own container / file-handle types, filled from the device field ----
    typedef DetSDW_SystemConfig SystemConfig;
    typedef DetSDW_SystemConfig_FileHandle SystemConfig_FileHandle;
    DetSDW_SystemConfig getCurrentSystemConfiguration() {
        arma::Cube<num> phi(pars_.N, pars_.opdim, pars_.m + 1);                // reference layout == ABI layout
        call(detsdw_get_phi(h_, phi.memptr()), "detsdw_get_phi");
        if (pars_.cdwU) {                   // src/detsdwopdim.cpp:5116-5122: the discrete field travels with phi
            std::vector<int32_t> l((size_t)pars_.N * (pars_.m + 1));
            call(detsdw_get_cdwl(h_, l.data()), "detsdw_get_cdwl");
            MatInt cdwl(pars_.N, pars_.m + 1);
            for (size_t i = 0; i < l.size(); ++i) cdwl[i] = l[i];
            return DetSDW_SystemConfig(pars_, phi, cdwl);
        }
        return DetSDW_SystemConfig(pars_, phi);
    }
    DetSDW_SystemConfig_FileHandle prepareSystemConfigurationStreamFileHandle(bool binaryStream, bool textStream,
                                                                              const std::string& directory = ".") {
        namespace fs = boost::filesystem;
        if (!(binaryStream || textStream)) throw_GeneralError("binaryStream or textStream must be sepcified to create file handle");
        DetSDW_SystemConfig_FileHandle fh;
        typedef DetSDW_SystemConfig_FileHandle::OfstreamPointer OfstreamPointer;
        if (binaryStream) {
            const fs::path p = fs::path(directory) / fs::path("configs-phi.binarystream");
            fh.phi_output_binary = OfstreamPointer(new std::ofstream(p.c_str(), std::ios::binary | std::ios::app));
            if (fh.phi_output_binary->fail()) std::cerr << "Could not open file " << p.string() << " for writing.\n";
        }
        if (textStream) {
            const fs::path p = fs::path(directory) / fs::path("configs-phi.textstream");
            fh.phi_output_text = OfstreamPointer(new std::ofstream(p.c_str(), std::ios::app));
            if (fh.phi_output_text->fail()) std::cerr << "Could not open file " << p.string() << " for writing.\n";
            fh.phi_output_text->precision(14);
            fh.phi_output_text->setf(std::ios::scientific, std::ios::floatfield);
        }
        if (pars_.cdwU) {                   // src/detsdwopdim.cpp:5158-5181
            if (binaryStream) {
                const fs::path p = fs::path(directory) / fs::path("configs-l.binarystream");
                fh.cdwl_output_binary = OfstreamPointer(new std::ofstream(p.c_str(), std::ios::binary | std::ios::app));
                if (fh.cdwl_output_binary->fail()) std::cerr << "Could not open file " << p.string() << " for writing.\n";
            }
            if (textStream) {
                const fs::path p = fs::path(directory) / fs::path("configs-l.textstream");
                fh.cdwl_output_text = OfstreamPointer(new std::ofstream(p.c_str(), std::ios::app));
                if (fh.cdwl_output_text->fail()) std::cerr << "Could not open file " << p.string() << " for writing.\n";
            }
        }
        return fh;
    }

    // ---- serialisation: DetQMC::saveContents / loadContents hand over their archive (src/detqmc.h:121-135).  The
    //      replica's state (fields, RNG stream position, step-size adaptation, update statistics) travels as one
    //      opaque blob in the format of detsdw_save_state. ----
    template<class Archive> void saveContents(Archive& ar) {
        TempFile tf;
        call(detsdw_save_state(h_, tf.path.c_str()), "detsdw_save_state");
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
        call(detsdw_load_state(h_, tf.path.c_str()), "detsdw_load_state");
        shadow_ = *rng_;                // the driver has just deserialised its wrapper: both streams are at the saved position
    }

private:
    struct TempFile {
        std::string path;
        TempFile() {
            char name[] = "/tmp/detsdwgpu-state-XXXXXX";
            const int fd = mkstemp(name);
            if (fd < 0) throw_GeneralError(std::string("mkstemp: ") + strerror(errno));
            close(fd);
            path = name;
        }
        ~TempFile() { std::remove(path.c_str()); }
    };
    // what RngWrapper::save hands to an archive (src/rngwrapper.h:100-105): seed, processIndex, state string
    struct RngProbe {
        std::vector<uint32_t> u;
        std::string state;
        RngProbe& operator<<(const uint32_t& v) { u.push_back(v); return *this; }
        RngProbe& operator<<(const std::string& s) { state = s; return *this; }
    };

    // ---- ONE random stream, two owners.  In the reference the replica draws from the RngWrapper its driver owns, and DetQMCPT
    //      takes the replica-exchange decisions of rank 0 from that same object (src/detqmcpt.h:1041).  The library owns a
    //      bit-identical copy of the stream (the device consumes pre-drawn windows of it), so the two are kept at the same
    //      position: before a sweep the library's stream skips what the driver drew in between, after a sweep the driver's
    //      wrapper (and a shadow copy that tells us where it was) skips what the replica consumed. ----
    static bool samePosition(const RngWrapper& a, const RngWrapper& b) {
        RngWrapper x = a, y = b;
        for (int i = 0; i < 3; ++i) if (x.rand01() != y.rand01()) return false;
        return true;
    }
    uint64_t libraryDrawn() const {
        detsdw_info i;
        call(detsdw_get_info(h_, &i), "detsdw_get_info");
        return i.rngDrawn;
    }
    void rngBeforeSweep() {
        int skipped = 0;
        while (!samePosition(*rng_, shadow_)) {                  // the driver drew from the wrapper since the last sweep
            (void)shadow_.rand01();
            (void)detsdw_rng_rand01(h_);
            if (++skipped > 100000) throw_GeneralError("DetSDWGpu: lost track of the shared random stream");
        }
        drawnBefore_ = libraryDrawn();
    }
    void rngAfterSweep() {
        const uint64_t used = libraryDrawn() - drawnBefore_;
        for (uint64_t i = 0; i < used; ++i) { (void)rng_->rand01(); (void)shadow_.rand01(); }
    }

    DetSDWGpu(RngWrapper& rng, const ModelParamsDetSDW& pars) : pars_(pars), rng_(&rng) {
        // The replica draws from ITS OWN copy of the stream (the device consumes pre-drawn windows of it).  DetQMC hands
        // over the RngWrapper it has just seeded with (rngSeed, simindex + 1) (src/detqmc.h:181): read the pair back
        // through the wrapper's serialisation hook and insist the stream is still at its start.
        RngProbe probe;
        boost::serialization::detail::member_saver<RngProbe, const RngWrapper>::invoke(probe, rng, 0);
        if (probe.u.size() != 2) throw_GeneralError("DetSDWGpu: unexpected RngWrapper layout");
        {
            RngWrapper fresh(probe.u[0], probe.u[1]);
            RngProbe p2;
            boost::serialization::detail::member_saver<RngProbe, const RngWrapper>::invoke(p2, fresh, 0);
            if (p2.state != probe.state) throw_GeneralError("DetSDWGpu: createReplica needs a freshly seeded RngWrapper");
        }
        if (probe.u[1] == 0) throw_GeneralError("DetSDWGpu: processIndex must be simindex + 1 >= 1");

        if (pars.turnoffFermions) throw_ParameterWrong_message("DetSDWGpu: turnoffFermions is not supported");
        if (pars.overRelaxation) throw_ParameterWrong_message("DetSDWGpu: overRelaxation is not supported");
        if (pars.phiFixed) throw_ParameterWrong_message("DetSDWGpu: phiFixed is not supported");
        if (pars.dumpGreensFunction) throw_ParameterWrong_message("DetSDWGpu: dumpGreensFunction is not supported");

        detsdw_params p;
        std::memset(&p, 0, sizeof(p));
        p.opdim = (int32_t)pars.opdim; p.L = (int32_t)pars.L; p.m = (int32_t)pars.m; p.s = (int32_t)pars.s;   // m, s: after updateTemperatureParameters
        p.delaySteps = (int32_t)pars.delaySteps;
        p.globalShift = pars.globalShift; p.globalUpdateInterval = (int32_t)pars.globalUpdateInterval;
        p.weakZflux = pars.weakZflux; p.phi2bosons = pars.phi2bosons;
        p.device = std::getenv("DQMC_DEVICE") ? std::atoi(std::getenv("DQMC_DEVICE")) : 0;
        p.simindex = (int32_t)probe.u[1] - 1; p.rngSeed = probe.u[0];
        p.has_mux_muy = 1;                                       // createReplica has already resolved mu -> mux, muy
        p.updateMethod = pars.updateMethod == ModelParamsDetSDW::ITERATIVE ? 0 : pars.updateMethod == ModelParamsDetSDW::WOODBURY ? 1 : 2;
        std::strncpy(p.bc, pars.bc_string.c_str(), sizeof(p.bc) - 1);
        p.beta = 0.0; p.dtau = pars.dtau;
        p.r = pars.r; p.c = pars.c; p.u = pars.u; p.lambda = pars.lambda;
        p.txhor = pars.txhor; p.txver = pars.txver; p.tyhor = pars.tyhor; p.tyver = pars.tyver;
        p.mu = pars.mu; p.mux = pars.mux; p.muy = pars.muy;
        p.accRatio = pars.accRatio; p.cdwU = pars.cdwU;
        // the UdV factorisation: "svd" as the reference's udvDecompose, "qr" (default) = same G to rounding, ~10x cheaper
        const char* stab = std::getenv("DQMC_STABILISATION");
        p.stabilisation = (stab && std::string(stab) == "svd") ? 0 : 1;
        p.cb_none = pars.checkerboard ? 0 : 1;
        p.wolffClusterUpdate = pars.wolffClusterUpdate; p.wolffClusterShiftUpdate = pars.wolffClusterShiftUpdate;
        p.repeatWolffPerSweep = (int32_t)pars.repeatWolffPerSweep;
        p.fermionMeasurements = pars.turnoffFermionMeasurements ? 0 : 1;
        p.spinProposalMethod = pars.spinProposalMethod == ModelParamsDetSDW::BOX ? 0 : pars.spinProposalMethod == ModelParamsDetSDW::ROTATE_THEN_SCALE ? 1 : 2;
        p.adaptScaleVariance = pars.adaptScaleVariance ? 1 : 0;
        p.repeatUpdateInSlice = (int32_t)pars.repeatUpdateInSlice;
        if (detsdw_create(&p, &h_) != DQMC_OK) throw_GeneralError(std::string("detsdw_create: ") + detsdw_last_error());
        // the constructor consumed the draws of setupRandomField: bring the driver's wrapper to the same position
        shadow_ = rng;
        drawnBefore_ = 0;
        rngAfterSweep();

        // the observables DetSDW registers (src/detsdwopdim.cpp:268-333), same names, same order
        using std::cref;
        obsScalar_.push_back(ScalarObservable(cref(normMeanPhi), "normMeanPhi", "nmp"));
        obsScalar_.push_back(ScalarObservable(cref(associatedEnergy), "associatedEnergy", ""));
        if (pars.opdim == 2) {
            obsScalar_.push_back(ScalarObservable(cref(phiRhoS_Gs), "phiRhoS_Gs", ""));
            obsScalar_.push_back(ScalarObservable(cref(phiRhoS_Gc), "phiRhoS_Gc", ""));
        }
        if (!pars.turnoffFermionMeasurements) {
            obsScalar_.push_back(ScalarObservable(cref(pairPlusMax), "pairPlusMax", "ppMax"));
            obsScalar_.push_back(ScalarObservable(cref(pairMinusMax), "pairMinusMax", "pmMax"));
            kOccX.zeros(pars.N); kOccY.zeros(pars.N); pairPlus.zeros(pars.N); pairMinus.zeros(pars.N);
            obsVector_.push_back(VectorObservable(cref(kOccX), pars.N, "kOccX", "nkx"));
            obsVector_.push_back(VectorObservable(cref(kOccY), pars.N, "kOccY", "nky"));
            obsScalar_.push_back(ScalarObservable(cref(greenK0), "greenK0", ""));
            obsScalar_.push_back(ScalarObservable(cref(greenLocal), "greenLocal", ""));
            obsVector_.push_back(VectorObservable(cref(pairPlus), pars.N, "pairPlus", "pp"));
            obsVector_.push_back(VectorObservable(cref(pairMinus), pars.N, "pairMinus", "pm"));
            obsScalar_.push_back(ScalarObservable(cref(occDiffSq), "occDiffSq", ""));
        }
    }

    void call(int rc, const char* what) const {
        if (rc != DQMC_OK) throw_GeneralError(std::string(what) + ": " + detsdw_last_error());
    }
    // after sweep(true): the members the observable handlers hold references to
    void refreshObservables() {
        detsdw_observables o;
        call(detsdw_get_observables(h_, &o), "detsdw_get_observables");
        normMeanPhi = o.normMeanPhi; associatedEnergy = o.associatedEnergy;
        phiRhoS_Gs = o.phiRhoS_Gs; phiRhoS_Gc = o.phiRhoS_Gc;
        if (o.fermionic_valid) {
            pairPlusMax = o.pairPlusMax; pairMinusMax = o.pairMinusMax;
            greenK0 = o.greenK0; greenLocal = o.greenLocal; occDiffSq = o.occDiffSq;
            call(detsdw_get_observable_vector(h_, DETSDW_OBS_KOCCX, kOccX.memptr()), "kOccX");
            call(detsdw_get_observable_vector(h_, DETSDW_OBS_KOCCY, kOccY.memptr()), "kOccY");
            call(detsdw_get_observable_vector(h_, DETSDW_OBS_PAIRPLUS, pairPlus.memptr()), "pairPlus");
            call(detsdw_get_observable_vector(h_, DETSDW_OBS_PAIRMINUS, pairMinus.memptr()), "pairMinus");
        }
    }
    static void writeHeader(const std::string& directory, const char* file, const std::string& info, const char* lastLine) {
        namespace fs = boost::filesystem;
        const fs::path p = fs::path(directory) / fs::path(file);
        if (fs::exists(p)) return;                                  // only if the file does not exist yet
        std::ofstream out(p.c_str(), std::ios::out);
        if (!out) { std::cerr << "Could not open file " << p.string() << " for writing.\n"; return; }
        out << info << lastLine;
    }

    ModelParamsDetSDW pars_;
    RngWrapper* rng_;                   // the driver's wrapper (DetQMC::rng / DetQMCPT::rng)
    RngWrapper shadow_;                 // where the wrapper was when we last looked
    uint64_t drawnBefore_ = 0;
    detsdw_replica* h_ = nullptr;
    std::vector<ScalarObservable> obsScalar_;
    std::vector<VectorObservable> obsVector_;
    std::vector<KeyValueObservable> obsKeyValue_;
    num normMeanPhi = 0, associatedEnergy = 0, phiRhoS_Gs = 0, phiRhoS_Gc = 0;
    num pairPlusMax = 0, pairMinusMax = 0, greenK0 = 0, greenLocal = 0, occDiffSq = 0;
    arma::Col<num> kOccX, kOccY, pairPlus, pairMinus;
};

// src/detsdwopdim.cpp:49-84 -- the same normalisation, with the reference's own functions
inline void createReplica(std::unique_ptr<DetSDWGpu>& replica_out, RngWrapper& rng, ModelParamsDetSDW pars,
                          DetModelLoggingParams loggingPars = DetModelLoggingParams(), const std::string& /*logfiledir*/ = "") {
    pars = updateTemperatureParameters(pars);
    pars.check();
    loggingPars.check();
    if (loggingPars.logSV || loggingPars.checkAndLogDetRatio || loggingPars.checkAndLogGreen || loggingPars.logGreenConsistency ||
        loggingPars.checkCheckerboardConsistency)
        throw_ParameterWrong_message("DetSDWGpu: the CPU self-check logs (logSV, checkAndLog*, ...) are not available");
    if (!(pars.specified.count("mux") && pars.specified.count("muy"))) { pars.mux = pars.mu; pars.muy = pars.mu; }
    replica_out = std::unique_ptr<DetSDWGpu>(new DetSDWGpu(rng, pars));
}

// get_replica_exchange_probability<Model> (src/detmodel.h:97-108) must be specialised per model; the reference does it for
// DetSDW<CB, OPDIM> (src/detsdwopdim.cpp:5251-5326, Hukushima & Nemoto 1996)
template<>
inline num get_replica_exchange_probability<DetSDWGpu>(num parameter_1, num action_contribution_1, num parameter_2, num action_contribution_2) {
    return detsdw_replica_exchange_probability(parameter_1, action_contribution_1, parameter_2, action_contribution_2);
}
