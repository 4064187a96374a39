// Host control flow of the Hubbard replica (see dethubbard.h).  Mirrors:
//   createReplica / updateTemperatureParameters / ModelParams<DetHubbard>::check   src/dethubbard.cpp:37-47, src/detmodelparams.h:68-122,
//                                                                                   src/dethubbardparams.cpp:21-55
//   DetHubbard ctor, setupRandomAuxfield                                            src/dethubbard.cpp:49-118, 690-700
//   sweep_skeleton / sweepDown / sweepUp (no global move in this model)             src/detmodel.h:1266-1478
//   initMeasurements / finishMeasurements                                           src/dethubbard.cpp:501-519, 637-649
// All numerics go through the C ABI in include/dqmc_hip.h.
#include "dethubbard.h"
#include <cmath>
#include <cstdio>
#include <cstring>

namespace detqmc {

void DetHubbard::check(int rc, const char* what) {
    if (rc != DQMC_OK) throw GeneralError(rc, std::string(what) + ": " + dqmc_last_error());
}

DetHubbard::DetHubbard(const dethubbard_params& in, int nchains) : p_(in) {
    if (nchains < 1) throw ParameterWrong("need at least one replica");
    // updateTemperatureParameters (detmodelparams.h:68-122)
    if (!(p_.dtau > 0)) throw ParameterWrong("Parameter dtau has incorrect value");
    if (p_.s <= 0) throw ParameterWrong("Parameter s has incorrect value");
    if (p_.beta > 0 && p_.m > 0) throw ParameterWrong("Only specify one of the parameters beta and m");
    if (!(p_.beta > 0) && p_.m <= 0) throw ParameterWrong("Specify either parameter m or beta");
    if (p_.m <= 0) p_.m = (int32_t)std::round(p_.beta / p_.dtau);
    p_.beta = p_.m * p_.dtau;
    while (p_.m <= p_.s) p_.s -= 1;
    if (p_.s < 1) throw ParameterWrong("Cannot choose parameter s obeying 0 < s < m");
    // ModelParams<DetHubbard>::check (dethubbardparams.cpp:21-55)
    if (p_.L <= 0) throw ParameterWrong("Parameter L has incorrect value");
    if (p_.d != 2) throw ParameterWrong("Hubbard replica: only d = 2 lattices are supported by this build");
    if (p_.checkerboard && p_.L % 2 != 0) throw ParameterWrong("Checker board decomposition only supported for even linear lattice sizes");
    if (p_.L % 2 != 0) throw ParameterWrong("this build needs an even linear lattice size");
    N_ = p_.L * p_.L; m_ = p_.m; s_ = p_.s; n_ = (m_ + s_ - 1) / s_;
    alpha_ = std::acosh(std::exp(p_.dtau * p_.U * 0.5));

    dqmc_params kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.model = DQMC_MODEL_HUBBARD;
    kp.opdim = 1; kp.L = p_.L; kp.m = m_; kp.s = s_; kp.delaySteps = 1; kp.bc = DQMC_BC_PBC; kp.device = p_.device;
    kp.stabilisation = p_.stabilisation; kp.cb_none = p_.checkerboard ? 0 : 1;
    kp.dtau = p_.dtau; kp.txhor = p_.t; kp.u = p_.U; kp.mux = p_.mu; kp.muy = p_.mu; kp.accRatio = 0.5;
    check(dqmc_create_batch(&kp, nchains, &ctx_), "dqmc_create");
    try {
        std::vector<double> all((size_t)nchains * (m_ + 1) * N_, 0.0);
        for (int b = 0; b < nchains; ++b) {
            ch_.emplace_back(p_.rngSeed, (uint32_t)(p_.simindex + b));
            Chain& c = ch_.back();
            c.aux.assign((size_t)(m_ + 1) * N_, 0.0);
            for (int k = 1; k <= m_; ++k)                                    // setupRandomAuxfield (:690-700)
                for (int site = 0; site < N_; ++site) c.aux[(size_t)k * N_ + site] = (c.rng.rand01() <= 0.5) ? +1.0 : -1.0;
            // slice 0 is never used; +1 keeps the cosh/sinh helper kernel away from 0/0
            for (int site = 0; site < N_; ++site) c.aux[site] = 1.0;
            std::memcpy(&all[(size_t)b * (m_ + 1) * N_], c.aux.data(), c.aux.size() * sizeof(double));
        }
        check(dqmc_set_fields_all_host(ctx_, all.data()), "dqmc_set_fields_all_host");
        check(dqmc_udv_setup(ctx_), "setupUdVStorage_and_calculateGreen");
    } catch (...) {
        dqmc_destroy(ctx_);
        throw;
    }
}

DetHubbard::~DetHubbard() { dqmc_destroy(ctx_); }

void DetHubbard::updateInSlice(int k) {
    check(dqmc_update_slice(ctx_, k, 0), "updateInSlice");
    if (measuring_) check(dqmc_measure_slice(ctx_), "measure");     // updateInSliceAndMaybeMeasure (detmodel.h:1279-1285)
}

void DetHubbard::sweepDown() {                                      // detmodel.h:1333-1399
    for (int k = m_; k >= (n_ - 1) * s_ + 1; --k) { updateInSlice(k); check(dqmc_wrap(ctx_, DQMC_DOWN, k), "wrapDownGreen"); }
    for (int l = n_ - 1; l >= 1; --l) {
        check(dqmc_advance(ctx_, DQMC_DOWN, l + 1), "advanceDownGreen");
        for (int k = l * s_; k >= (l - 1) * s_ + 1; --k) { updateInSlice(k); check(dqmc_wrap(ctx_, DQMC_DOWN, k), "wrapDownGreen"); }
    }
    check(dqmc_advance(ctx_, DQMC_DOWN, 1), "advanceDownGreen");
}

void DetHubbard::sweepUp() {                                        // detmodel.h:1266-1325
    check(dqmc_reset_storage0(ctx_), "reset storage[0]");
    for (int l = 0; l <= n_ - 2; ++l) {
        for (int k = l * s_ + 1; k <= (l + 1) * s_; ++k) { check(dqmc_wrap(ctx_, DQMC_UP, k - 1), "wrapUpGreen"); updateInSlice(k); }
        check(dqmc_advance(ctx_, DQMC_UP, l), "advanceUpGreen");
    }
    for (int k = (n_ - 1) * s_ + 1; k <= m_; ++k) { check(dqmc_wrap(ctx_, DQMC_UP, k - 1), "wrapUpGreen"); updateInSlice(k); }
    check(dqmc_advance(ctx_, DQMC_UP, n_ - 1), "advanceUpGreen");
}

void DetHubbard::sweep_skeleton(bool takeMeasurements) {
    const size_t need = (size_t)2 * N_ * m_;                        // per proposal: one draw for the site, at most one for Metropolis
    const size_t nb = ch_.size();
    window_.resize(need * nb);
    for (size_t b = 0; b < nb; ++b) std::memcpy(&window_[b * need], ch_[b].rng.peek(need), need * sizeof(double));
    check(dqmc_push_uniforms_all_host(ctx_, window_.data(), need), "dqmc_push_uniforms_all_host");
    if (takeMeasurements) check(dqmc_measure_reset(ctx_), "initMeasurements");
    measuring_ = takeMeasurements;
    try {
        if (lastSweepDir_ == Up) sweepDown(); else sweepUp();
    } catch (...) { measuring_ = false; throw; }
    measuring_ = false;
    lastSweepDir_ = (lastSweepDir_ == Up) ? Down : Up;
    ++performedSweeps_;
    std::vector<dqmc_update_state> st(nb);
    check(dqmc_get_update_states_all_host(ctx_, st.data()), "dqmc_get_update_states_all_host");
    for (size_t b = 0; b < nb; ++b) {
        ch_[b].rng.consume((size_t)st[b].rng_consumed);
        ch_[b].lastAccRatio = st[b].lastAccRatio;
        if (takeMeasurements) finishMeasurements((int)b); else ch_[b].obs.valid = 0;
    }
}

void DetHubbard::sweep(bool takeMeasurements) { sweep_skeleton(takeMeasurements); }

// finishMeasurements (dethubbard.cpp:637-649) from the device accumulators (layout: kernels_hubbard.hip)
void DetHubbard::finishMeasurements(int b) {
    check(dqmc_select_chain(ctx_, b), "dqmc_select_chain");
    std::vector<double> acc(dqmc_measure_accum_size(ctx_));
    check(dqmc_measure_read_host(ctx_, acc.data()), "dqmc_measure_read_host");
    if ((int)acc[5] != m_) throw GeneralError(DQMC_EINVAL, "measurement sweep did not visit every time slice");
    const double N = N_, m = m_;
    dethubbard_observables& o = ch_[b].obs;
    o.occUp = 1.0 - (1.0 / (N * m)) * acc[0];
    o.occDn = 1.0 - (1.0 / (N * m)) * acc[1];
    o.occTotal = o.occUp + o.occDn;
    o.occDouble = 1.0 + (1.0 / (N * m)) * (acc[4] - acc[0] - acc[1]);
    o.localMoment = o.occTotal - 2 * o.occDouble;
    o.ePotential = p_.U * o.occDouble;
    o.eKinetic = (p_.t / (N * m)) * (acc[2] + acc[3]) - p_.mu * o.occTotal;
    o.eTotal = o.eKinetic + o.ePotential;
    o.valid = 1;
    ch_[b].zcorr.assign(N_, 0.0);
    for (int j = 0; j < N_; ++j) ch_[b].zcorr[j] = acc[6 + j] / m;
}

void DetHubbard::getZcorr(double* out, int b) const {
    if (!ch_[b].obs.valid) throw GeneralError(DQMC_EINVAL, "no measurement has been taken");
    std::memcpy(out, ch_[b].zcorr.data(), ch_[b].zcorr.size() * sizeof(double));
}

void DetHubbard::getInfo(dethubbard_info& o, int b) {
    std::memset(&o, 0, sizeof(o));
    o.L = p_.L; o.N = N_; o.m = m_; o.s = s_; o.n = n_; o.performedSweeps = performedSweeps_; o.lastSweepDir = (int)lastSweepDir_;
    o.currentTimeslice = dqmc_current_timeslice(ctx_);
    o.beta = p_.beta; o.dtau = p_.dtau; o.alpha = alpha_; o.lastAccRatio = ch_[b].lastAccRatio; o.rngDrawn = ch_[b].rng.drawn();
}

void DetHubbard::getAuxfield(double* out, int b) {
    check(dqmc_select_chain(ctx_, b), "dqmc_select_chain");
    check(dqmc_get_fields_host(ctx_, ch_[b].aux.data(), nullptr, nullptr), "dqmc_get_fields_host");
    // device layout [m+1][N] == the reference's auxfield(N, m+1) column-major
    std::memcpy(out, ch_[b].aux.data(), ch_[b].aux.size() * sizeof(double));
}

void DetHubbard::getGreen(double* gUp, double* gDn, int b) {
    check(dqmc_select_chain(ctx_, b), "dqmc_select_chain");
    const int ng = 2 * N_;
    std::vector<dqmc_cplx> g((size_t)ng * ng);
    check(dqmc_get_green_host(ctx_, g.data()), "dqmc_get_green_host");
    for (int j = 0; j < N_; ++j)
        for (int i = 0; i < N_; ++i) {
            if (gUp) gUp[(size_t)j * N_ + i] = g[(size_t)j * ng + i].re;
            if (gDn) gDn[(size_t)j * N_ + i] = g[(size_t)(N_ + j) * ng + (N_ + i)].re;
        }
}

// ---- checkpoint / resume ----
namespace {
const char kHubMagic[8] = {'D', 'Q', 'M', 'C', 'H', 'U', 'B', '1'};
struct HubFile { std::FILE* f; ~HubFile() { if (f) std::fclose(f); } };
void hwr(std::FILE* f, const void* p, size_t n) { if (std::fwrite(p, 1, n, f) != n) throw GeneralError(DQMC_EINVAL, "checkpoint: write error"); }
void hrd(std::FILE* f, void* p, size_t n) { if (std::fread(p, 1, n, f) != n) throw GeneralError(DQMC_EINVAL, "checkpoint: truncated file"); }
}  // namespace

void DetHubbard::saveState(const std::string& path) {
    HubFile hf{std::fopen(path.c_str(), "wb")};
    if (!hf.f) throw GeneralError(DQMC_EINVAL, "Could not open file " + path + " for writing");
    const int32_t hdr[6] = {(int32_t)ch_.size(), p_.L, m_, s_, performedSweeps_, (int32_t)lastSweepDir_};
    hwr(hf.f, kHubMagic, 8); hwr(hf.f, hdr, sizeof(hdr));
    for (int b = 0; b < (int)ch_.size(); ++b) {
        Chain& c = ch_[b];
        check(dqmc_select_chain(ctx_, b), "dqmc_select_chain");
        check(dqmc_get_fields_host(ctx_, c.aux.data(), nullptr, nullptr), "dqmc_get_fields_host");
        const std::vector<uint64_t> rng = c.rng.serialize();
        const uint64_t nr = rng.size(), na = c.aux.size();
        hwr(hf.f, &nr, 8); hwr(hf.f, rng.data(), nr * 8);
        hwr(hf.f, &na, 8); hwr(hf.f, c.aux.data(), na * 8);
    }
}

void DetHubbard::loadState(const std::string& path) {
    HubFile hf{std::fopen(path.c_str(), "rb")};
    if (!hf.f) throw GeneralError(DQMC_EINVAL, "Could not open file " + path + " for reading");
    char magic[8]; int32_t hdr[6];
    hrd(hf.f, magic, 8); hrd(hf.f, hdr, sizeof(hdr));
    if (std::memcmp(magic, kHubMagic, 8) != 0) throw GeneralError(DQMC_EINVAL, "checkpoint: not a detqmc_amd Hubbard state file");
    if (hdr[0] != (int32_t)ch_.size() || hdr[1] != p_.L || hdr[2] != m_ || hdr[3] != s_)
        throw GeneralError(DQMC_EINVAL, "checkpoint: written for a different replica set-up");
    std::vector<double> all(ch_.size() * (size_t)(m_ + 1) * N_);
    for (size_t b = 0; b < ch_.size(); ++b) {
        Chain& c = ch_[b];
        uint64_t nr = 0, na = 0;
        hrd(hf.f, &nr, 8);
        if (nr < DSFMT19937::state_words() + 3 || nr > (1u << 26)) throw GeneralError(DQMC_EINVAL, "checkpoint: bad RNG record");
        std::vector<uint64_t> rng(nr);
        hrd(hf.f, rng.data(), nr * 8);
        hrd(hf.f, &na, 8);
        if (na != c.aux.size()) throw GeneralError(DQMC_EINVAL, "checkpoint: field size mismatch");
        hrd(hf.f, c.aux.data(), na * 8);
        c.rng.deserialize(rng);
        std::memcpy(&all[b * c.aux.size()], c.aux.data(), na * 8);
    }
    check(dqmc_set_fields_all_host(ctx_, all.data()), "dqmc_set_fields_all_host");
    check(dqmc_udv_setup(ctx_), "setupUdVStorage_and_calculateGreen");       // dethubbard.h:345-350
    performedSweeps_ = hdr[4];
    lastSweepDir_ = Up;
}

}  // namespace detqmc

// ---------------------------------------------------------------------------------------------
// C API (include/dethubbard_host.h)
// ---------------------------------------------------------------------------------------------
using detqmc::DetHubbard;
struct dethubbard_replica { DetHubbard* impl; int sel; };
static thread_local std::string g_hub_err;
#define HGUARD(...)                                                                                \
    if (!r) { g_hub_err = "null replica handle"; return DQMC_EINVAL; }                             \
    try { __VA_ARGS__; return DQMC_OK; }                                                           \
    catch (const detqmc::GeneralError& e) { g_hub_err = e.what(); return e.code; }                 \
    catch (const std::exception& e) { g_hub_err = e.what(); return DQMC_EINVAL; }

extern "C" const char* dethubbard_last_error(void) { return g_hub_err.c_str(); }
extern "C" int dethubbard_create(const dethubbard_params* p, int nchains, dethubbard_replica** out) {
    if (!p || !out) { g_hub_err = "null argument"; return DQMC_EINVAL; }
    *out = nullptr;
    try { DetHubbard* d = new DetHubbard(*p, nchains); *out = new dethubbard_replica{d, 0}; return DQMC_OK; }
    catch (const detqmc::GeneralError& e) { g_hub_err = e.what(); return e.code; }
    catch (const std::exception& e) { g_hub_err = e.what(); return DQMC_EINVAL; }
}
extern "C" void dethubbard_destroy(dethubbard_replica* r) { if (r) { delete r->impl; delete r; } }
extern "C" int dethubbard_select_chain(dethubbard_replica* r, int chain) {
    if (!r || chain < 0 || chain >= r->impl->numChains()) { g_hub_err = "chain index out of range"; return DQMC_EINVAL; }
    r->sel = chain;
    return DQMC_OK;
}
extern "C" int dethubbard_sweep(dethubbard_replica* r, int tm) { HGUARD(r->impl->sweep(tm != 0)) }
extern "C" int dethubbard_sweep_thermalization(dethubbard_replica* r) { HGUARD(r->impl->sweepThermalization()) }
extern "C" int dethubbard_get_info(dethubbard_replica* r, dethubbard_info* out) { HGUARD(r->impl->getInfo(*out, r->sel)) }
extern "C" int dethubbard_get_auxfield(dethubbard_replica* r, double* out) { HGUARD(r->impl->getAuxfield(out, r->sel)) }
extern "C" int dethubbard_get_green(dethubbard_replica* r, double* gUp, double* gDn) { HGUARD(r->impl->getGreen(gUp, gDn, r->sel)) }
extern "C" int dethubbard_get_observables(dethubbard_replica* r, dethubbard_observables* out) { HGUARD(r->impl->getObservables(*out, r->sel)) }
extern "C" int dethubbard_get_zcorr(dethubbard_replica* r, double* out) { HGUARD(r->impl->getZcorr(out, r->sel)) }
extern "C" int dethubbard_save_state(dethubbard_replica* r, const char* path) {
    if (!path) { g_hub_err = "null path"; return DQMC_EINVAL; }
    HGUARD(r->impl->saveState(path))
}
extern "C" int dethubbard_load_state(dethubbard_replica* r, const char* path) {
    if (!path) { g_hub_err = "null path"; return DQMC_EINVAL; }
    HGUARD(r->impl->loadState(path))
}
extern "C" double dethubbard_rng_rand01(dethubbard_replica* r) { return r ? r->impl->rand01(r->sel) : -1.0; }
extern "C" dqmc_ctx* dethubbard_ctx(dethubbard_replica* r) { return r ? r->impl->ctx() : nullptr; }
