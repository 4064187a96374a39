// TEST INFRASTRUCTURE ONLY -- fixture generator that drives the *real* reference Hubbard replica
// (crstnbr/detqmc src/dethubbard.{h,cpp}, read-only under /root/reference; BASELINE config 1).  Compiled from the
// reference sources where they lie (Makefile target `hubbard`); nothing of the reference is copied.
//
// Usage: ref_harness_hubbard <outdir> key=value ...      (L, d, beta, dtau, s, t, U, mu, checkerboard, sweeps, measureSweeps,
//                                                          rngSeed, simindex)
// Reference entry points exercised (file:line in /root/reference/src):
//   createReplica / DetHubbard ctor, setupRandomAuxfield     dethubbard.cpp:37-47, 49-118, 690-700
//   setupPropTmat_direct / _checkerboard                      dethubbard.cpp:702-770
//   computeBmat, weightRatioSingleFlip, updateGreenFunctionWithFlip, updateInSlice   dethubbard.cpp:772-849, 141-172
//   sweep / sweepThermalization                               dethubbard.cpp:921-945 -> detmodel.h:1408-1478
//   measure / finishMeasurements                              dethubbard.cpp:521-577, 637-649
#include <iostream>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <armadillo>
#include "boost/serialization/string.hpp"
#include "boost/serialization/set.hpp"
#include "boost/serialization/vector.hpp"
#include "boost/assign/std/vector.hpp"
#define private public
#define protected public
#include "rngwrapper.h"
#include "detmodel.h"
#include "dethubbard.h"
#undef private
#undef protected

static std::string g_outdir;
static std::ofstream g_manifest;
static void dump_raw(const std::string& name, const void* data, size_t bytes, const std::vector<size_t>& shape) {
    std::string fn = g_outdir + "/" + name + ".bin";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) { perror(fn.c_str()); exit(2); }
    fwrite(data, 1, bytes, f);
    fclose(f);
    g_manifest << name << " f8";
    for (size_t s : shape) g_manifest << " " << s;
    g_manifest << "\n";
}
static void dump(const std::string& name, const arma::Mat<double>& m) { dump_raw(name, m.memptr(), m.n_elem * 8, {m.n_rows, m.n_cols}); }
static void dump(const std::string& name, const arma::Col<double>& m) { dump_raw(name, m.memptr(), m.n_elem * 8, {m.n_elem}); }
static void dump_scalar(const std::string& name, double v) { dump_raw(name, &v, 8, {1}); }

template<class T> static T get(const std::map<std::string, std::string>& kv, const std::string& k, T def) {
    auto it = kv.find(k);
    if (it == kv.end()) return def;
    std::istringstream ss(it->second);
    T v; ss >> v; return v;
}

static void dump_state(DetHubbard& rep, const std::string& tag) {
    dump(tag + "_auxfield", arma::conv_to<arma::Mat<double>>::from(rep.auxfield));
    dump(tag + "_gUp", rep.gUp);
    dump(tag + "_gDn", rep.gDn);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: ref_harness_hubbard <outdir> key=value...\n"; return 2; }
    g_outdir = argv[1];
    std::map<std::string, std::string> kv;
    for (int i = 2; i < argc; ++i) { std::string a(argv[i]); size_t eq = a.find('='); kv[a.substr(0, eq)] = a.substr(eq + 1); }
    g_manifest.open(g_outdir + "/manifest.txt");
    uint32_t seed = get<uint32_t>(kv, "rngSeed", 1020304050u), simindex = get<uint32_t>(kv, "simindex", 0);
    RngWrapper rng(seed, simindex + 1);
    ModelParams<DetHubbard> p;
#define SET(field, type, def) p.field = get<type>(kv, #field, def); p.specified.insert(#field);
    SET(L, uint32_t, 4) SET(d, uint32_t, 2) SET(beta, double, 2.0) SET(dtau, double, 0.1) SET(s, uint32_t, 10)
    SET(t, double, 1.0) SET(U, double, 4.0) SET(mu, double, 0.0)
#undef SET
    p.checkerboard = get<int>(kv, "checkerboard", 0) != 0; p.specified.insert("checkerboard");
    try {
        std::unique_ptr<DetHubbard> rep;
        createReplica(rep, rng, p, DetModelLoggingParams());
        arma::Col<double> meta(6);
        meta[0] = p.L; meta[1] = rep->N; meta[2] = rep->m; meta[3] = rep->s; meta[4] = rep->n; meta[5] = rep->alpha;
        dump("meta", meta);
        dump("proptmat", rep->proptmat);
        dump_state(*rep, "init");
        {
            arma::Mat<double> dall(rep->N, rep->n + 1);
            for (uint32_t l = 0; l <= rep->n; ++l) dall.col(l) = rep->UdVStorageUp[l].d;
            dump("init_udv_d_up", dall);
        }
        dump("bmat_up_k3", rep->computeBmat(3, 2, DetHubbard::Spin::Up));
        dump("bmat_dn_chain", rep->computeBmat(std::min<uint32_t>(rep->s, rep->m), 0, DetHubbard::Spin::Down));
        uint32_t nsweeps = get<uint32_t>(kv, "sweeps", 2);
        for (uint32_t i = 1; i <= nsweeps; ++i) {
            rep->sweepThermalization();
            dump_state(*rep, "sweep" + std::to_string(i));
        }
        uint32_t nmeas = get<uint32_t>(kv, "measureSweeps", 0);
        for (uint32_t i = 1; i <= nmeas; ++i) {
            rep->sweep(true);
            std::string tag = "meas" + std::to_string(i);
            dump_state(*rep, tag);
            arma::Col<double> o(8);
            o[0] = rep->occUp; o[1] = rep->occDn; o[2] = rep->occTotal; o[3] = rep->occDouble; o[4] = rep->localMoment;
            o[5] = rep->eKinetic; o[6] = rep->ePotential; o[7] = rep->eTotal;
            dump(tag + "_obs", o);
            dump(tag + "_zcorr", rep->zcorr);
        }
        arma::Col<double> nxt(4);
        for (int i = 0; i < 4; ++i) nxt[i] = rng.rand01();
        dump("rng_next", nxt);
    } catch (const std::exception& e) {
        std::cerr << "reference threw: " << e.what() << "\n";
        return 3;
    }
    return 0;
}
