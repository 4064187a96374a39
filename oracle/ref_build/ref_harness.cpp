// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product.
//
// Fixture generator + CPU timer that drives the *real* reference implementation
// (crstnbr/detqmc, read-only under /root/reference) for the DetSDW hot path.  It is compiled
// from the reference sources where they lie (see Makefile in this directory); nothing of the
// reference is copied into this repository.  Output: raw little-endian arrays plus a manifest,
// packed into tests/golden/*.npz by oracle/make_golden.py.
//
// Usage: ref_harness <outdir> key=value ...
//   mode=dump   : write fixtures for one parameter set
//   mode=time   : run sweeps and print sweeps/s (CPU baseline, kind "reference")
//   mode=timeparts : time the slices of the top block and one stabilisation step (CPU baseline at sizes where a sweep takes hours)
//
// Reference entry points exercised (file:line in /root/reference/src):
//   createReplica                      detsdwopdim.cpp:49-84
//   DetSDW ctor / setupRandomField     detsdwopdim.cpp:158-361, 1099-1113
//   leftMultiplyBk & friends           detsdwopdim.cpp:1996-2420
//   computeBmatSDW                     detsdwopdim.cpp:1309-1497
//   updateInSlice / _delayed           detsdwopdim.cpp:2428-2489, 3023-3175
//   sweepThermalization / sweep        detsdwopdim.cpp:4423-4502 -> detmodel.h:1408-1478
//   globalMove                         detsdwopdim.cpp:3461-3641
//   RngWrapper                         rngwrapper.h:42-62, rngwrapper.cpp:31-49

// Pull in everything third-party FIRST with normal access control ...
#include <iostream>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <set>
#include <list>
#include <tuple>
#include <memory>
#include <complex>
#include <functional>
#include <algorithm>
#include <numeric>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <armadillo>
#include "boost/serialization/string.hpp"
#include "boost/serialization/set.hpp"
#include "boost/serialization/list.hpp"
#include "boost/serialization/vector.hpp"
#include "boost/serialization/export.hpp"
#include "boost/serialization/split_member.hpp"
#include "boost/archive/binary_oarchive.hpp"
#include "boost/archive/binary_iarchive.hpp"
#include "boost/assign/std/vector.hpp"
#include "boost/filesystem.hpp"
#include "boost/timer/timer.hpp"
// ... then open up the reference's own classes so the harness can read protected state
// (SURVEY.md section 8c).  Class layout is unaffected by access specifiers with g++.
#define private public
#define protected public
#include "rngwrapper.h"
#include "detmodel.h"
#include "detsdwopdim.h"
#undef private
#undef protected

typedef std::complex<double> cpx_t;
#ifndef HARNESS_OPDIM
#define HARNESS_OPDIM 2
#endif
// one binary per order-parameter dimension, like the reference's own detqmcsdwo{1,2,3} targets
typedef DetSDW<CB_ASSAAD_BERG, HARNESS_OPDIM> SDWN;
typedef DetSDW<CB_NONE, HARNESS_OPDIM> SDWN_DENSE;     // checkerboard=false: dense B = e^{-dtau V} e^{-dtau K}

static std::string g_outdir;
static std::ofstream g_manifest;

static void dump_raw(const std::string& name, const char* dtype, const void* data, size_t bytes,
                     const std::vector<size_t>& shape) {
    std::string fn = g_outdir + "/" + name + ".bin";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) { perror(fn.c_str()); exit(2); }
    fwrite(data, 1, bytes, f);
    fclose(f);
    g_manifest << name << " " << dtype;
    for (size_t s : shape) g_manifest << " " << s;
    g_manifest << "\n";
}
// all matrices are written column-major (Fortran order); the manifest shape is (rows, cols[, slices])
static void dump(const std::string& name, const arma::Mat<cpx_t>& m) {
    dump_raw(name, "c16", m.memptr(), m.n_elem * sizeof(cpx_t), {m.n_rows, m.n_cols});
}
static void dump(const std::string& name, const arma::Mat<double>& m) {
    dump_raw(name, "f8", m.memptr(), m.n_elem * sizeof(double), {m.n_rows, m.n_cols});
}
static void dump(const std::string& name, const arma::Cube<double>& m) {
    dump_raw(name, "f8", m.memptr(), m.n_elem * sizeof(double), {m.n_rows, m.n_cols, m.n_slices});
}
static void dump(const std::string& name, const arma::Col<double>& m) {
    dump_raw(name, "f8", m.memptr(), m.n_elem * sizeof(double), {m.n_elem});
}
static void dump_scalar(const std::string& name, double v) {
    dump_raw(name, "f8", &v, sizeof(double), {1});
}

static std::map<std::string, std::string> parse_args(int argc, char** argv) {
    std::map<std::string, std::string> kv;
    for (int i = 2; i < argc; ++i) {
        std::string a(argv[i]);
        size_t eq = a.find('=');
        if (eq == std::string::npos) { std::cerr << "bad arg " << a << "\n"; exit(2); }
        kv[a.substr(0, eq)] = a.substr(eq + 1);
    }
    return kv;
}

template<class T> static T get(const std::map<std::string, std::string>& kv, const std::string& k, T def) {
    auto it = kv.find(k);
    if (it == kv.end()) return def;
    std::istringstream ss(it->second);
    T v; ss >> v; return v;
}
static std::string gets(const std::map<std::string, std::string>& kv, const std::string& k, const std::string& def) {
    auto it = kv.find(k);
    return it == kv.end() ? def : it->second;
}

static ModelParamsDetSDW make_params(const std::map<std::string, std::string>& kv) {
    ModelParamsDetSDW p;
#define SET(field, type, def) p.field = get<type>(kv, #field, def); p.specified.insert(#field);
    SET(opdim, uint32_t, 2)
    SET(L, uint32_t, 4)
    SET(beta, double, 2.0)
    SET(dtau, double, 0.1)
    SET(s, uint32_t, 10)
    SET(r, double, -1.0)
    SET(c, double, 3.0)
    SET(u, double, 1.0)
    SET(lambda, double, 1.0)
    SET(txhor, double, -1.0)
    SET(txver, double, -0.5)
    SET(tyhor, double, 0.5)
    SET(tyver, double, 1.0)
    SET(mu, double, -0.5)
    SET(accRatio, double, 0.5)
    SET(delaySteps, uint32_t, 16)
    SET(globalUpdateInterval, uint32_t, 100)
    SET(repeatUpdateInSlice, uint32_t, 1)
    SET(cdwU, double, 0.0)
#undef SET
    if (kv.count("mux")) { p.mux = get<double>(kv, "mux", 0.0); p.specified.insert("mux"); }
    if (kv.count("muy")) { p.muy = get<double>(kv, "muy", 0.0); p.specified.insert("muy"); }
    p.checkerboard = get<int>(kv, "checkerboard", 1) != 0; p.specified.insert("checkerboard");
    p.updateMethod_string = gets(kv, "updateMethod", "delayed"); p.specified.insert("updateMethod");
    p.spinProposalMethod_string = gets(kv, "spinProposalMethod", "box"); p.specified.insert("spinProposalMethod");
    p.adaptScaleVariance = get<int>(kv, "adaptScaleVariance", 0) != 0; p.specified.insert("adaptScaleVariance");
    p.bc_string = gets(kv, "bc", "pbc"); p.specified.insert("bc");
    p.weakZflux = get<int>(kv, "weakZflux", 0) != 0; p.specified.insert("weakZflux");
    p.globalShift = get<int>(kv, "globalShift", 0) != 0; p.specified.insert("globalShift");
    p.wolffClusterUpdate = get<int>(kv, "wolffClusterUpdate", 0) != 0; p.specified.insert("wolffClusterUpdate");
    p.wolffClusterShiftUpdate = get<int>(kv, "wolffClusterShiftUpdate", 0) != 0; p.specified.insert("wolffClusterShiftUpdate");
    p.repeatWolffPerSweep = get<uint32_t>(kv, "repeatWolffPerSweep", 1);
    p.turnoffFermionMeasurements = get<int>(kv, "fermionMeas", 0) == 0; p.specified.insert("turnoffFermionMeasurements");
    p.phiFixed = get<int>(kv, "phiFixed", 0) != 0;
    return p;
}

// deterministic, asymmetric, well-scaled test matrix both sides can regenerate
static arma::Mat<cpx_t> test_matrix(uint32_t n) {
    arma::Mat<cpx_t> A(n, n);
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t i = 0; i < n; ++i)
            A(i, j) = cpx_t(std::sin(0.37 * i + 1.31 * j + 0.11 * i * j), std::cos(0.73 * i - 0.29 * j + 0.05 * i * j));
    return A;
}

template<class SDW>
static void dump_state(SDW& rep, const std::string& tag) {
    dump(tag + "_phi", rep.phi);
    if (rep.pars.cdwU) dump(tag + "_cdwl", arma::conv_to<arma::Mat<double>>::from(rep.cdwl));   // discrete field l_i(tau_k), N x (m+1)
    dump(tag + "_g", rep.g);
    dump(tag + "_g_inv_sv", rep.g_inv_sv);
    dump_scalar(tag + "_phiDelta", rep.ad.phiDelta);
    dump_scalar(tag + "_lastAccRatio", rep.ad.lastAccRatioLocal_phi);
    if (rep.pars.spinProposalMethod_string != "box") {        // rotate / scale proposals (detsdwopdim.cpp:3934-4170, adaptation :3299-3375)
        dump_scalar(tag + "_angleDelta", rep.ad.angleDelta);
        dump_scalar(tag + "_scaleDelta", rep.ad.scaleDelta);
    }
    dump_scalar(tag + "_accGlobalShifts", rep.us.acceptedGlobalShifts);
    dump_scalar(tag + "_attGlobalShifts", rep.us.attemptedGlobalShifts);
    dump_scalar(tag + "_accWolff", rep.us.acceptedWolffClusterUpdates);
    dump_scalar(tag + "_attWolff", rep.us.attemptedWolffClusterUpdates);
    dump_scalar(tag + "_accWolffShift", rep.us.acceptedWolffClusterShiftUpdates);
    dump_scalar(tag + "_attWolffShift", rep.us.attemptedWolffClusterShiftUpdates);
    dump_scalar(tag + "_addedWolffClusterSize", rep.us.addedWolffClusterSize);
}

template<class SDW>
static int run(const std::map<std::string, std::string>& kv) {
    std::string mode = gets(kv, "mode", "dump");
    uint32_t seed = get<uint32_t>(kv, "rngSeed", 1020304050u);
    uint32_t simindex = get<uint32_t>(kv, "simindex", 0);
    RngWrapper rng(seed, simindex + 1);   // detqmc.h:181

    ModelParamsDetSDW pars = make_params(kv);
    std::unique_ptr<SDW> rep;
    createReplica(rep, rng, pars, DetModelLoggingParams(), g_outdir);

    const uint32_t m = rep->m, n = rep->n, s = rep->s;
    const uint32_t ng = rep->sz;

    if (mode == "time") {
        uint32_t warm = get<uint32_t>(kv, "warmup", 2);
        uint32_t sweeps = get<uint32_t>(kv, "sweeps", 4);
        for (uint32_t i = 0; i < warm; ++i) rep->sweepThermalization();
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0; i < sweeps; ++i) rep->sweepThermalization();
        auto t1 = std::chrono::steady_clock::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        printf("REF_TIMING sweeps=%u seconds=%.6f sweeps_per_s=%.6f\n", sweeps, sec, sweeps / sec);
        return 0;
    }

    if (mode == "timeparts") {
        // CPU baseline for sizes where a whole reference sweep takes hours (n_g = 2304): the slices of the top partial block and
        // ONE stabilisation step (advanceDownGreen through greenFromUdV: the general case) of a down sweep, timed separately; the
        // caller extrapolates  sweep = m * slice + n * advance.  Run it at a small beta (m >= 3) -- the per-slice and per-step
        // costs depend on the lattice, not on beta.
        typename SDW::sdwLeftMultiplyBmatInv lInv(rep.get());
        typename SDW::sdwRightMultiplyBmat rB(rep.get());
        auto t0 = std::chrono::steady_clock::now();
        uint32_t nsl = 0;
        for (uint32_t k = m; k >= (n - 1) * s + 1; --k) {
            rep->updateInSliceThermalization(k);
            rep->wrapDownGreen(lInv, rB, k, 0);
            ++nsl;
        }
        auto t1 = std::chrono::steady_clock::now();
        rep->advanceDownGreen(rB, n, 0);
        auto t2 = std::chrono::steady_clock::now();
        printf("REF_PARTS slices=%u slice_seconds=%.6f advance_seconds=%.6f m=%u n=%u s=%u ng=%u\n", nsl,
               std::chrono::duration<double>(t1 - t0).count() / nsl, std::chrono::duration<double>(t2 - t1).count(), m, n, s, ng);
        return 0;
    }
    {
        arma::Col<double> meta(8);
        meta[0] = pars.opdim; meta[1] = pars.L; meta[2] = m; meta[3] = s; meta[4] = n; meta[5] = ng;
        meta[6] = seed; meta[7] = simindex;
        dump("meta", meta);
    }

    // --- state right after construction: random field, caches, UdV storage, G(beta) ---
    dump("init_phi", rep->phi);
    if (pars.cdwU) dump("init_cdwl", arma::conv_to<arma::Mat<double>>::from(rep->cdwl));
    dump("init_coshTermPhi", rep->coshTermPhi);
    dump("init_sinhTermPhi", rep->sinhTermPhi);
    dump("init_g", rep->g);
    dump("init_g_inv_sv", rep->g_inv_sv);
    {
        arma::Mat<double> dall(ng, n + 1);
        for (uint32_t l = 0; l <= n; ++l) dall.col(l) = (*rep->UdVStorage)[0][l].d;
        dump("init_udv_d", dall);
        if (get<int>(kv, "dumpUdV", 0)) {
            for (uint32_t l = 0; l <= n; ++l) {
                dump("init_udv_U_" + std::to_string(l), (*rep->UdVStorage)[0][l].U);
                dump("init_udv_Vt_" + std::to_string(l), (*rep->UdVStorage)[0][l].V_t);
            }
        }
    }

    // --- the four B multipliers on a fixed test matrix, single slice and chains ---
    if (!get<int>(kv, "setupOnly", 0)) {      // setupOnly=1: only the state after construction (largest sizes)
        arma::Mat<cpx_t> A = test_matrix(ng);
        uint32_t k = std::min<uint32_t>(3, m);
        dump_scalar("bmult_k", k);
        typename SDW::sdwLeftMultiplyBmat lB(rep.get());
        typename SDW::sdwRightMultiplyBmat rB(rep.get());
        typename SDW::sdwLeftMultiplyBmatInv lBi(rep.get());
        typename SDW::sdwRightMultiplyBmatInv rBi(rep.get());
        dump("bmult_left",     lB(0, A, k, k - 1));
        dump("bmult_leftinv",  lBi(0, A, k, k - 1));
        dump("bmult_right",    rB(0, A, k, k - 1));
        dump("bmult_rightinv", rBi(0, A, k, k - 1));
        uint32_t k2 = std::min<uint32_t>(s, m), k1 = 0;
        dump_scalar("bchain_k2", k2);
        dump("bchain_left",     lB(0, A, k2, k1));
        dump("bchain_leftinv",  lBi(0, A, k2, k1));
        dump("bchain_right",    rB(0, A, k2, k1));
        dump("bchain_rightinv", rBi(0, A, k2, k1));
        // dense B (computeBmatSDW) -- differs from the checkerboard product at O(dtau^2)
        dump("bdense_k", rep->computeBmatSDW(k, k - 1));
    }

    // --- one slice of local updates at k=m starting from G(beta) (a17/a18/a20) ---
    uint32_t nsweeps = get<uint32_t>(kv, "sweeps", 2);
    if (get<int>(kv, "sliceTrace", 1)) {
        rep->updateInSliceThermalization(m);   // detsdwopdim.cpp:3294 (calls updateInSlice + adaptation)
        dump("slice_phi_m", arma::Mat<double>(rep->phi.slice(m)));
        if (pars.cdwU) dump("slice_cdwl_m", arma::conv_to<arma::Mat<double>>::from(rep->cdwl.col(m)));
        dump("slice_g", rep->g);
        dump_scalar("slice_accRatio", rep->ad.lastAccRatioLocal_phi);
        // finish that down-sweep by hand exactly as sweepDown would (detmodel.h:1364-1392)
        typename SDW::sdwLeftMultiplyBmatInv lInv(rep.get());
        typename SDW::sdwRightMultiplyBmat rB(rep.get());
        rep->wrapDownGreen(lInv, rB, m, 0);
        dump("slice_g_wrapped", rep->g);
        // sliceTrace=2: this one slice only (the largest sizes: a full CPU sweep at n_g = 2304 takes hours)
        if (get<int>(kv, "sliceTrace", 1) == 2) { dump_scalar("slice_phiDelta", rep->ad.phiDelta); return 0; }
        for (uint32_t k = m - 1; k >= (n - 1) * s + 1; --k) {
            rep->updateInSliceThermalization(k);
            rep->wrapDownGreen(lInv, rB, k, 0);
        }
        for (uint32_t l = n - 1; l >= 1; --l) {
            rep->advanceDownGreen(rB, l + 1, 0);
            if (l == n - 1) {
                dump("adv_g", rep->g);
                dump("adv_g_inv_sv", rep->g_inv_sv);
            }
            for (uint32_t k = l * s; k >= (l - 1) * s + 1; --k) {
                rep->updateInSliceThermalization(k);
                rep->wrapDownGreen(lInv, rB, k, 0);
            }
        }
        rep->advanceDownGreen(rB, 1, 0);
        rep->lastSweepDir = SDW::SweepDirection::Down;
        ++rep->performedSweeps;
        dump_state(*rep, "sweep1");
        for (uint32_t i = 2; i <= nsweeps; ++i) {
            rep->sweepThermalization();
            dump_state(*rep, "sweep" + std::to_string(i));
        }
    } else {
        for (uint32_t i = 1; i <= nsweeps; ++i) {
            rep->sweepThermalization();
            dump_state(*rep, "sweep" + std::to_string(i));
        }
    }
    // measurement sweeps: sweep(true) with the bosonic observables of measure()/finishMeasurements()
    // (detsdwopdim.cpp:509-545, :903-921); the harness runs with turnoffFermionMeasurements
    {
        uint32_t nmeas = get<uint32_t>(kv, "measureSweeps", 0);
        for (uint32_t i = 1; i <= nmeas; ++i) {
            rep->sweep(true);
            std::string tag = "meas" + std::to_string(i);
            dump_state(*rep, tag);
            arma::Col<double> mp(HARNESS_OPDIM);
            for (int d = 0; d < HARNESS_OPDIM; ++d) mp[d] = rep->meanPhi[d];
            dump(tag + "_meanPhi", mp);
            dump_scalar(tag + "_normMeanPhi", rep->normMeanPhi);
            dump_scalar(tag + "_associatedEnergy", rep->associatedEnergy);
            dump_scalar(tag + "_phiRhoS_Gc", rep->phiRhoS_Gc);
            dump_scalar(tag + "_phiRhoS_Gs", rep->phiRhoS_Gs);
            if (get<int>(kv, "fermionMeas", 0)) {     // measure() :545-899, finishMeasurements() :923-1017
                dump_scalar(tag + "_greenK0", rep->greenK0);
                dump_scalar(tag + "_greenLocal", rep->greenLocal);
                dump(tag + "_kOccX", rep->kOccX);
                dump(tag + "_kOccY", rep->kOccY);
                dump(tag + "_pairPlus", rep->pairPlus);
                dump(tag + "_pairMinus", rep->pairMinus);
                dump_scalar(tag + "_pairPlusMax", rep->pairPlusMax);
                dump_scalar(tag + "_pairMinusMax", rep->pairMinusMax);
                dump_scalar(tag + "_occDiffSq", rep->occDiffSq);
            }
        }
        if (get<int>(kv, "fermionMeas", 0)) dump("final_shiftGreenSymmetric", rep->shiftGreenSymmetric());
    }
    dump_scalar("exchange_action", rep->get_exchange_action_contribution());
    // on-disk configuration stream (detsdwopdim.cpp:4991-5012): the reference appends to <dir>/configs-phi.binarystream
    if (get<int>(kv, "cfgStream", 0)) {
        rep->saveConfigurationStreamBinary(g_outdir);
        rep->saveConfigurationStreamBinary(g_outdir);      // twice: the file is opened in append mode
    }
    // what the generator hands out next: pins the number of draws consumed so far
    {
        arma::Col<double> nxt(4);
        for (int i = 0; i < 4; ++i) nxt[i] = rng.rand01();
        dump("rng_next", nxt);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: ref_harness <outdir> key=value...\n"; return 2; }
    g_outdir = argv[1];
    auto kv = parse_args(argc, argv);
    std::string mode = gets(kv, "mode", "dump");

    if (mode == "rng") {
        // golden vector for the RNG restatements: (seed, processIndex) -> first numbers of rand01()
        g_manifest.open(g_outdir + "/manifest.txt");
        uint32_t seeds[3][2] = {{1020304050u, 1}, {5555u, 1}, {1020304050u, 2}};
        for (auto& sp : seeds) {
            RngWrapper rng(sp[0], sp[1]);
            arma::Col<double> v(2000);
            for (int i = 0; i < 2000; ++i) v[i] = rng.rand01();
            dump("rng_" + std::to_string(sp[0]) + "_" + std::to_string(sp[1]), v);
        }
        return 0;
    }
    if (mode != "time" && mode != "timeparts") g_manifest.open(g_outdir + "/manifest.txt");

    uint32_t opdim = get<uint32_t>(kv, "opdim", 2);
    try {
        if (opdim == HARNESS_OPDIM) return get<int>(kv, "checkerboard", 1) ? run<SDWN>(kv) : run<SDWN_DENSE>(kv);
    } catch (const std::exception& e) {
        std::cerr << "reference threw: " << e.what() << "\n";
        return 3;
    }
    std::cerr << "bad opdim\n";
    return 2;
}
