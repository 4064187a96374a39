// Host control flow of one SDW replica (see detsdw.h).  Mirrors, function by function:
//   createReplica / updateTemperatureParameters / ModelParamsDetSDW::check
//       src/detsdwopdim.cpp:49-84, src/detmodelparams.h:68-122, src/detsdwparams.cpp:21-140
//   DetSDW ctor, setupRandomField                      src/detsdwopdim.cpp:158-361, 1099-1113
//   sweep_skeleton / sweepThermalization_skeleton      src/detmodel.h:1408-1478
//   sweepDown / sweepUp                                src/detmodel.h:1333-1399 / 1266-1325
//   globalMove / attemptGlobalShiftMove / phiAction    src/detsdwopdim.cpp:3461-3486, 3565-3644, 4242-4300
// All numerics go through the C ABI in include/dqmc_hip.h.
#include "detsdw.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <thread>

namespace detqmc {

void DetSDW::check(int rc, const char* what) {
    if (rc != DQMC_OK) throw GeneralError(rc, std::string(what) + ": " + dqmc_last_error());
}

// field-wise comparison of two normalised parameter sets (struct padding is not the caller's business); everything but
// the exchange parameter r, the device and -- unless seeds_too -- the RNG stream identity
static bool same_model(const detsdw_params& a, const detsdw_params& b, bool seeds_too) {
    const bool ints = a.opdim == b.opdim && a.L == b.L && a.m == b.m && a.s == b.s && a.delaySteps == b.delaySteps &&
                      a.globalShift == b.globalShift && a.globalUpdateInterval == b.globalUpdateInterval &&
                      a.weakZflux == b.weakZflux && a.phi2bosons == b.phi2bosons && a.has_mux_muy == b.has_mux_muy &&
                      a.updateMethod == b.updateMethod && a.stabilisation == b.stabilisation && a.cb_none == b.cb_none &&
                      a.wolffClusterUpdate == b.wolffClusterUpdate && a.wolffClusterShiftUpdate == b.wolffClusterShiftUpdate &&
                      a.repeatWolffPerSweep == b.repeatWolffPerSweep && a.fermionMeasurements == b.fermionMeasurements &&
                      a.spinProposalMethod == b.spinProposalMethod && a.adaptScaleVariance == b.adaptScaleVariance &&
                      a.repeatUpdateInSlice == b.repeatUpdateInSlice;
    const bool reals = a.beta == b.beta && a.dtau == b.dtau && a.c == b.c && a.u == b.u && a.lambda == b.lambda &&
                       a.txhor == b.txhor && a.txver == b.txver && a.tyhor == b.tyhor && a.tyver == b.tyver &&
                       a.mu == b.mu && a.mux == b.mux && a.muy == b.muy && a.accRatio == b.accRatio && a.cdwU == b.cdwU;
    const bool bc = std::strncmp(a.bc[0] ? a.bc : "pbc", b.bc[0] ? b.bc : "pbc", sizeof(a.bc)) == 0;
    const bool seeds = !seeds_too || (a.rngSeed == b.rngSeed && a.simindex == b.simindex);
    return ints && reals && bc && seeds;
}

dqmc_ctx* DetSDW::select(int b) {
    if (b < 0 || b >= (int)ch_.size()) throw ParameterWrong("chain index out of range");
    Group& g = grp(b);
    check(dqmc_select_chain(g.ctx, b - g.first), "dqmc_select_chain");
    return g.ctx;
}

// fn(group) for every sub-batch; with more than one group each runs on its own host thread (contexts are independent:
// own stream, own device buffers; the kernel ABI allows different contexts on different threads).  The first exception
// is rethrown after all threads have finished.
void DetSDW::forEachGroup(const std::function<void(Group&)>& fn) {
    if (groups_.size() == 1) { fn(groups_[0]); return; }
    std::vector<std::exception_ptr> err(groups_.size());
    std::vector<std::thread> th;
    th.reserve(groups_.size());
    for (size_t i = 0; i < groups_.size(); ++i)
        th.emplace_back([&, i]() { try { fn(groups_[i]); } catch (...) { err[i] = std::current_exception(); } });
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

// updateTemperatureParameters + ModelParamsDetSDW::check + createReplica on one parameter set
void DetSDW::normalise(detsdw_params& p, int& bcv) {
    // --- updateTemperatureParameters (detmodelparams.h:68-122) ---
    if (!(p.dtau > 0)) throw ParameterWrong("Parameter dtau has incorrect value");
    if (p.s <= 0) throw ParameterWrong("Parameter s has incorrect value");
    if (p.beta > 0 && p.m > 0) throw ParameterWrong("Only specify one of the parameters beta and m");
    if (!(p.beta > 0) && p.m <= 0) throw ParameterWrong("Specify either parameter m or beta");
    if (p.m > 0) {
        p.beta = p.m * p.dtau;
    } else {
        p.m = (int32_t)std::round(p.beta / p.dtau);
        p.beta = p.m * p.dtau;
    }
    while (p.m <= p.s) p.s -= 1;
    if (p.s < 1) throw ParameterWrong("Cannot choose parameter s obeying 0 < s < m");
    // --- ModelParamsDetSDW::check (detsdwparams.cpp:21-140) ---
    if (!(p.opdim == 1 || p.opdim == 2 || p.opdim == 3)) throw ParameterWrong("Parameter opdim has incorrect value");
    const std::string bc(p.bc[0] ? p.bc : "pbc");
    if (bc == "pbc") bcv = DQMC_BC_PBC;
    else if (bc == "apbc-x") bcv = DQMC_BC_APBC_X;
    else if (bc == "apbc-y") bcv = DQMC_BC_APBC_Y;
    else if (bc == "apbc-xy") bcv = DQMC_BC_APBC_XY;
    else throw ParameterWrong("Parameter bc has incorrect value: " + bc);
    if (p.weakZflux && p.opdim != 2)
        throw ParameterWrong("Magnetic field specified for opdim=" + std::to_string(p.opdim) +
                             ", but currently only supported for opdim=2");
    if (p.updateMethod < 0 || p.updateMethod > 2) throw ParameterWrong("Parameter updateMethod has incorrect value");
    const int N = p.L * p.L;
    if (p.updateMethod == 2 && (p.delaySteps <= 0 || p.delaySteps > N))
        throw ParameterWrong("Parameter delaySteps has incorrect value");
    if (p.repeatWolffPerSweep == 0) p.repeatWolffPerSweep = 1;
    if (p.repeatUpdateInSlice == 0) p.repeatUpdateInSlice = 1;
    if (p.repeatUpdateInSlice < 1) throw ParameterWrong("Parameter repeatUpdateInSlice has incorrect value");
    if (p.spinProposalMethod < 0 || p.spinProposalMethod > 2) throw ParameterWrong("Parameter spinProposalMethod has incorrect value");     // detsdwparams.cpp:81-87
    if (p.spinProposalMethod != 0 && p.opdim != 3)      // the reference throws from the first proposal (detsdwopdim.cpp:3938, 4012, 4085)
        throw ParameterWrong("spinProposalMethod rotate_then_scale / rotate_and_scale is only supported for the O(3) model");
    if ((p.globalShift || p.wolffClusterUpdate || p.wolffClusterShiftUpdate) && p.globalUpdateInterval == 0)
        throw ParameterWrong("Parameter globalUpdateInterval has incorrect value");                      // detsdwparams.cpp:89-93
    if (p.wolffClusterShiftUpdate && (p.globalShift || p.wolffClusterUpdate))
        throw ParameterWrong("Either use combined wolffClusterShiftUpdate or individual global updates");   // :94-96
    if (p.repeatWolffPerSweep < 1) throw ParameterWrong("Parameter repeatWolffPerSweep has incorrect value");
    if (p.L % 2 != 0) throw ParameterWrong("Checker board decomposition only supported for even linear lattice sizes");
    if (p.stabilisation != 0 && p.stabilisation != 1) throw ParameterWrong("Parameter stabilisation has incorrect value");
    // createReplica (detsdwopdim.cpp:75-79)
    if (!p.has_mux_muy) { p.mux = p.mu; p.muy = p.mu; }
}

DetSDW::DetSDW(const detsdw_params* in, int nchains, int sub_batches) {
    if (!in || nchains < 1) throw ParameterWrong("need at least one replica");
    int S = sub_batches;
    if (S == 0) {                                 // automatic: up to 4 groups, each at least 32 chains
        S = 1;
        for (int cand = 4; cand >= 2; --cand)
            if (nchains % cand == 0 && nchains / cand >= 32) { S = cand; break; }
    }
    if (S < 1 || nchains % S != 0) throw ParameterWrong("sub_batches must divide the number of replicas");
    int bcv = 0;
    for (int b = 0; b < nchains; ++b) {
        detsdw_params p = in[b];
        normalise(p, bcv);
        if (b > 0) {
            // the chains of a batch are the replicas of ONE parallel-tempering ensemble: same lattice, same
            // temperature and couplings; they may differ in r (the exchange parameter) and in the RNG stream
            if (!same_model(ch_[0].pars, p, /*seeds_too=*/false))
                throw ParameterWrong("replicas of one batch may differ only in r, rngSeed and simindex");
        }
        ch_.emplace_back(p);
    }
    const detsdw_params& p = ch_[0].pars;
    const int N = p.L * p.L;
    N_ = N; opdim_ = p.opdim; MSF_ = (p.opdim == 3) ? 4 : 2; ng_ = MSF_ * N_;
    m_ = p.m; s_ = p.s; n_ = (m_ + s_ - 1) / s_;

    dqmc_params kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.opdim = p.opdim; kp.L = p.L; kp.m = p.m; kp.s = p.s;
    // iterative / woodbury update the Green's function after every accepted proposal: D = 1
    kp.delaySteps = (p.updateMethod == 2) ? p.delaySteps : 1;
    kp.bc = bcv; kp.weakZflux = p.weakZflux; kp.phi2bosons = p.phi2bosons; kp.device = p.device;
    kp.dtau = p.dtau; kp.r = p.r; kp.c = p.c; kp.u = p.u; kp.lambda = p.lambda;
    kp.txhor = p.txhor; kp.txver = p.txver; kp.tyhor = p.tyhor; kp.tyver = p.tyver;
    kp.mux = p.mux; kp.muy = p.muy; kp.accRatio = p.accRatio; kp.cdwU = p.cdwU;
    kp.stabilisation = p.stabilisation;
    kp.cb_none = p.cb_none ? 1 : 0;          // reference option checkerboard=false (DetSDW<CB_NONE, OPDIM>)
    kp.rng_window_per_site = uniformsPerSite();
    // result-neutral execution choices.  The pipelined update pays only while few contexts share the GPU (with more of them the
    // contexts overlap each other instead): automatic here means at most two sub-batches.
    kp.tuning = p.tuning;
    if (kp.tuning.pipeline == 0 && S > 2) kp.tuning.pipeline = -1;
    groups_.resize(S);
    try {
        for (int g = 0; g < S; ++g) {
            groups_[g].first = g * (nchains / S);
            groups_[g].count = nchains / S;
            check(dqmc_create_batch(&kp, nchains / S, &groups_[g].ctx), "dqmc_create");
        }
        for (int b = 0; b < nchains; ++b) {
            Chain& c = ch_[b];
            c.phi.assign((size_t)N_ * opdim_ * (m_ + 1), 0.0);
            c.cdwl.assign((size_t)N_ * (m_ + 1), 0);
            setupRandomField(c);
            dqmc_ctx* ctx = select(b);
            check(dqmc_set_exchange_parameter(ctx, c.pars.r), "dqmc_set_exchange_parameter");
            check(dqmc_set_fields_host(ctx, c.phi.data()), "dqmc_set_fields_host");
            if (c.pars.cdwU != 0.0) check(dqmc_set_cdwl_host(ctx, c.cdwl.data()), "dqmc_set_cdwl_host");
        }
        forEachGroup([this](Group& g) { setupUdVStorage_and_calculateGreen(g); });
    } catch (...) {
        for (auto& g : groups_) dqmc_destroy(g.ctx);
        throw;
    }
    lastSweepDir_ = Up;                                    // detmodel.h:711
}

DetSDW::~DetSDW() { for (auto& g : groups_) dqmc_destroy(g.ctx); }

// detsdwopdim.cpp:1099-1113: k outer, site, dim; one more draw per site for the discrete field (drawn whatever cdwU is)
void DetSDW::setupRandomField(Chain& c) {
    for (int k = 1; k <= m_; ++k)
        for (int site = 0; site < N_; ++site) {
            for (int dim = 0; dim < opdim_; ++dim) c.phi[phiIdx(site, dim, k)] = c.rng.randRange(-1.0, 1.0);
            const double r = c.rng.rand01();
            c.cdwl[(size_t)k * N_ + site] = (r <= 0.25) ? +2 : (r <= 0.5) ? -2 : (r <= 0.75) ? +1 : -1;
        }
}

void DetSDW::setupUdVStorage_and_calculateGreen(Group& g) {
    check(dqmc_udv_setup(g.ctx), "setupUdVStorage_and_calculateGreen");
}

// Ship the worst-case number of upcoming uniforms of this sweep; the device consumes a prefix.
void DetSDW::beginLocalUpdates(Group& g) {
    const size_t need = (size_t)uniformsPerSite() * N_ * m_;
    g.window.resize(need * g.count);               // all chains' windows back to back: one host -> device transfer
    for (int b = 0; b < g.count; ++b) std::memcpy(&g.window[b * need], ch_[g.first + b].rng.peek(need), need * sizeof(double));
    check(dqmc_push_uniforms_all_host(g.ctx, g.window.data(), need), "dqmc_push_uniforms_all_host");
}
void DetSDW::endLocalUpdates(Group& g) {
    std::vector<dqmc_update_state> st(g.count);
    check(dqmc_get_update_states_all_host(g.ctx, st.data()), "dqmc_get_update_states_all_host");
    for (int b = 0; b < g.count; ++b) {
        Chain& c = ch_[g.first + b];
        c.rng.consume((size_t)st[b].rng_consumed);
        c.phiDelta = st[b].phiDelta;
        c.lastAccRatio = st[b].lastAccRatio;
        c.angleDelta = st[b].angleDelta;
        c.scaleDelta = st[b].scaleDelta;
    }
}

// per site and slice of a sweep: what the local updates can consume from the RNG stream -- box: opdim proposal draws + at most one
// acceptance draw; rotate / scale: a Gaussian draw costs a variable number (polar Box-Muller), 8 is far above the mean of ~ 3.6 and an
// overrun is reported (DQMC_ERNG), never silent; times repeatUpdateInSlice; the cdwl pass: one proposal + one acceptance draw
int DetSDW::uniformsPerSite() const {
    const detsdw_params& p = ch_[0].pars;
    const int per_pass = p.spinProposalMethod == 0 ? p.opdim + 1 : 8;
    return per_pass * p.repeatUpdateInSlice + (p.cdwU != 0.0 ? 2 : 0);
}

// which proposal / adaptation this sweep uses (updateInSlice, updateInSliceThermalization: detsdwopdim.cpp:2438-2470, 3299-3321)
void DetSDW::updateInSlice(Group& g, int k, bool thermalization) {
    const detsdw_params& p = ch_[0].pars;
    int proposal = DQMC_PROPOSE_BOX, adapt = DQMC_ADAPT_BOX;
    if (p.spinProposalMethod == 1) {                       // rotate_then_scale: each sweep alternates between rotating and scaling
        proposal = (performedSweeps_ % 2 == 0) ? DQMC_PROPOSE_ROTATE : DQMC_PROPOSE_SCALE;
        adapt = (performedSweeps_ % 2 == 0) ? DQMC_ADAPT_ROTATE : DQMC_ADAPT_SCALE;
    } else if (p.spinProposalMethod == 2) {                // rotate_and_scale: the adapted quantity alternates every 100 sweeps
        proposal = DQMC_PROPOSE_ROTATE_AND_SCALE;
        adapt = (performedSweeps_ % 200 < 100) ? DQMC_ADAPT_ROTATE : DQMC_ADAPT_SCALE;
    }
    check(dqmc_update_slice_ex(g.ctx, k, thermalization ? 1 : 0, proposal, adapt, p.adaptScaleVariance, p.repeatUpdateInSlice), "updateInSlice");
    if (measuring_) check(dqmc_measure_slice(g.ctx), "measure");     // updateInSliceAndMaybeMeasure (detmodel.h:1279-1285)
}

// detmodel.h:1333-1399
void DetSDW::sweepDown(Group& g, bool thermalization) {
    dqmc_ctx* ctx_ = g.ctx;
    for (int k = m_; k >= (n_ - 1) * s_ + 1; --k) {
        updateInSlice(g, k, thermalization);
        check(dqmc_wrap(ctx_, DQMC_DOWN, k), "wrapDownGreen");
    }
    for (int l = n_ - 1; l >= 1; --l) {
        check(dqmc_advance(ctx_, DQMC_DOWN, l + 1), "advanceDownGreen");
        for (int k = l * s_; k >= (l - 1) * s_ + 1; --k) {
            updateInSlice(g, k, thermalization);
            check(dqmc_wrap(ctx_, DQMC_DOWN, k), "wrapDownGreen");
        }
    }
    check(dqmc_advance(ctx_, DQMC_DOWN, 1), "advanceDownGreen");
}

// detmodel.h:1266-1325
void DetSDW::sweepUp(Group& g, bool thermalization) {
    dqmc_ctx* ctx_ = g.ctx;
    check(dqmc_reset_storage0(ctx_), "reset storage[0]");
    for (int l = 0; l <= n_ - 2; ++l) {
        for (int k = l * s_ + 1; k <= (l + 1) * s_; ++k) {
            check(dqmc_wrap(ctx_, DQMC_UP, k - 1), "wrapUpGreen");
            updateInSlice(g, k, thermalization);
        }
        check(dqmc_advance(ctx_, DQMC_UP, l), "advanceUpGreen");
    }
    for (int k = (n_ - 1) * s_ + 1; k <= m_; ++k) {
        check(dqmc_wrap(ctx_, DQMC_UP, k - 1), "wrapUpGreen");
        updateInSlice(g, k, thermalization);
    }
    check(dqmc_advance(ctx_, DQMC_UP, n_ - 1), "advanceUpGreen");
}

// detmodel.h:1408-1478 and detsdwopdim.cpp:4423-4502
void DetSDW::sweep_skeleton(Group& g, bool thermalization) {
    if (lastSweepDir_ == Up) {
        globalMove(g);
        beginLocalUpdates(g);
        sweepDown(g, thermalization);
        endLocalUpdates(g);
    } else {
        beginLocalUpdates(g);
        sweepUp(g, thermalization);
        endLocalUpdates(g);
    }
}

// sweep(takeMeasurements): the bosonic observables depend on the field only, and measure(k) runs right after the
// updates of slice k (updateInSliceAndMaybeMeasure, detmodel.h:1279-1285, 1346-1352) -- no slice changes again within
// the sweep, so they are accumulated afterwards from the final field, in the slice order of the sweep just done.
void DetSDW::sweep(bool takeMeasurements) {
    const bool fermionic = takeMeasurements && ch_[0].pars.fermionMeasurements;
    if (fermionic) for (auto& g : groups_) check(dqmc_measure_reset(g.ctx), "initMeasurements");
    measuring_ = fermionic;
    try { forEachGroup([this](Group& g) { sweep_skeleton(g, false); }); } catch (...) { measuring_ = false; throw; }
    measuring_ = false;
    lastSweepDir_ = (lastSweepDir_ == Up) ? Down : Up;
    ++performedSweeps_;
    for (int b = 0; b < (int)ch_.size(); ++b) {
        if (takeMeasurements) {
            syncPhiFromDevice(b);
            measureBosonic(ch_[b], lastSweepDir_ == Down);
            if (fermionic) finishFermionic(b);
        } else {
            ch_[b].obs.valid = 0;
            ch_[b].obs.fermionic_valid = 0;
        }
    }
}

// finishMeasurements, fermionic part (detsdwopdim.cpp:923-1015) from the device accumulators of chain b
void DetSDW::finishFermionic(int b) {
    Chain& c = ch_[b];
    dqmc_ctx* ctx_ = select(b);
    std::vector<double> acc(dqmc_measure_accum_size(ctx_));
    check(dqmc_measure_read_host(ctx_, acc.data()), "dqmc_measure_read_host");
    const int L = c.pars.L, N = N_, m = m_;
    if ((int)acc[3] != m) throw GeneralError(DQMC_EINVAL, "measurement sweep did not visit every time slice");
    detsdw_observables& o = c.obs;
    o.greenK0 = acc[0] / double(m);
    o.greenLocal = acc[1] / double(m);
    o.occDiffSq = acc[2] / double(m);
    c.pairPlus.assign(N, 0.0); c.pairMinus.assign(N, 0.0); c.kOccX.assign(N, 0.0); c.kOccY.assign(N, 0.0);
    for (int i = 0; i < N; ++i) { c.pairPlus[i] = acc[4 + i] / m; c.pairMinus[i] = acc[4 + N + i] / m; }
    // momentum-space occupation: Fourier sum over the (2L-1)^2 site-difference bins, k offset by half a step along
    // antiperiodic directions (:616-659)
    const int W = 2 * L - 1, nbins = W * W;
    const double* S[2] = {&acc[4 + 2 * N], &acc[4 + 2 * N + 2 * (size_t)nbins]};
    const std::string bc(c.pars.bc[0] ? c.pars.bc : "pbc");
    const double offx = (bc == "apbc-x" || bc == "apbc-xy") ? 0.5 : 0.0, offy = (bc == "apbc-y" || bc == "apbc-xy") ? 0.5 : 0.0;
    const double pi = M_PI;
    // e^{i (kx dx + ky dy)} = e^{i kx dx} e^{i ky dy}: two small tables instead of a cos/sin per (k, bin)
    std::vector<double> ex((size_t)L * W * 2), ey((size_t)L * W * 2);
    for (int kk = 0; kk < L; ++kk) {
        const double kx = -pi + (double(kk) + offx) * 2 * pi / double(L), ky = -pi + (double(kk) + offy) * 2 * pi / double(L);
        for (int d = 0; d < W; ++d) {
            const int dd = d - (L - 1);
            ex[((size_t)kk * W + d) * 2] = std::cos(kx * dd); ex[((size_t)kk * W + d) * 2 + 1] = std::sin(kx * dd);
            ey[((size_t)kk * W + d) * 2] = std::cos(ky * dd); ey[((size_t)kk * W + d) * 2 + 1] = std::sin(ky * dd);
        }
    }
    for (int ksite = 0; ksite < N; ++ksite) {
        const double* tx = &ex[(size_t)(ksite % L) * W * 2];
        const double* ty = &ey[(size_t)(ksite / L) * W * 2];
        double sx = 0.0, sy = 0.0;
        for (int bin = 0; bin < nbins; ++bin) {
            const int ix = bin % W, iy = bin / W;
            const double cs = tx[2 * ix] * ty[2 * iy] - tx[2 * ix + 1] * ty[2 * iy + 1];
            const double sn = tx[2 * ix] * ty[2 * iy + 1] + tx[2 * ix + 1] * ty[2 * iy];
            sx += cs * S[0][2 * bin] - sn * S[0][2 * bin + 1];
            sy += cs * S[1][2 * bin] - sn * S[1][2 * bin + 1];
        }
        c.kOccX[ksite] = 2.0 - sx / double(m * N);      // 2.0: spin included (:938-941)
        c.kOccY[ksite] = 2.0 - sy / double(m * N);
    }
    // pairing correlations at maximum distance: the 3 x 3 sites around (L/2, L/2) (:986-1002)
    double pp = 0.0, pm = 0.0;
    for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox) {
            const int i = (L / 2 + oy) * L + (L / 2 + ox);
            pp += c.pairPlus[i]; pm += c.pairMinus[i];
        }
    o.pairPlusMax = pp / 9.0;
    o.pairMinusMax = pm / 9.0;
    o.fermionic_valid = 1;
}

void DetSDW::getObservableVector(int which, double* out, int b) const {
    const Chain& c = ch_[b];
    if (!c.obs.fermionic_valid) throw GeneralError(DQMC_EINVAL, "no fermionic measurement has been taken");
    const std::vector<double>* v = which == DETSDW_OBS_KOCCX ? &c.kOccX : which == DETSDW_OBS_KOCCY ? &c.kOccY
                                 : which == DETSDW_OBS_PAIRPLUS ? &c.pairPlus : which == DETSDW_OBS_PAIRMINUS ? &c.pairMinus : nullptr;
    if (!v) throw ParameterWrong("unknown observable vector");
    std::memcpy(out, v->data(), v->size() * sizeof(double));
}

// initMeasurements / measure / finishMeasurements, bosonic part (detsdwopdim.cpp:441-456, :509-545, :903-921)
void DetSDW::measureBosonic(Chain& c, bool descending) {
    const int L = c.pars.L;
    double meanPhi[3] = {0.0, 0.0, 0.0};
    double Gc = 0.0, Gs = 0.0, assoc = 0.0;
    auto dot = [&](const double* a, const double* bb) {       // arma::dot on short vectors: two accumulators
        double v1 = 0.0, v2 = 0.0;
        int i = 0;
        for (; i + 1 < opdim_; i += 2) { v1 += a[i] * bb[i]; v2 += a[i + 1] * bb[i + 1]; }
        if (i < opdim_) v1 += a[i] * bb[i];
        return v1 + v2;
    };
    for (int kk = 0; kk < m_; ++kk) {
        const int k = descending ? m_ - kk : 1 + kk;
        auto get = [&](int site, double* out) { for (int d = 0; d < opdim_; ++d) out[d] = c.phi[phiIdx(site, d, k)]; };
        if (opdim_ == 2) {
            for (int site = 0; site < N_; ++site) {
                const int x = site % L, y = site / L;
                double ps[3], px[3], py[3];
                get(site, ps); get(y * L + (x + 1) % L, px); get(((y + 1) % L) * L + x, py);
                Gc += dot(ps, px) + dot(ps, py);
                Gs += px[0] * ps[1] - px[1] * ps[0];
            }
        }
        for (int site = 0; site < N_; ++site) {
            double ps[3];
            get(site, ps);
            for (int d = 0; d < opdim_; ++d) meanPhi[d] += ps[d];
            assoc += dot(ps, ps);
        }
    }
    detsdw_observables& o = c.obs;
    std::memset(&o, 0, sizeof(o));
    double nrm2 = 0.0;   // (fermionic fields are filled in afterwards by finishFermionic)
    for (int d = 0; d < opdim_; ++d) { o.meanPhi[d] = meanPhi[d] / double(N_ * m_); nrm2 += o.meanPhi[d] * o.meanPhi[d]; }
    o.normMeanPhi = std::sqrt(nrm2);
    if (opdim_ == 2) { o.phiRhoS_Gc = Gc * (0.5 * c.pars.dtau); o.phiRhoS_Gs = Gs * c.pars.dtau; }
    o.associatedEnergy = assoc / (2.0 * N_ * m_);
    o.valid = 1;
}
void DetSDW::sweepThermalization() {
    forEachGroup([this](Group& g) { sweep_skeleton(g, true); });
    lastSweepDir_ = (lastSweepDir_ == Up) ? Down : Up;
    ++performedSweeps_;
}

// detsdwopdim.cpp:3461-3486 -- all chains of a batch attempt their global moves in the same sweep, in the reference's
// order: shift, Wolff cluster, combined cluster + shift
void DetSDW::globalMove(Group& g) {
    const detsdw_params& p = ch_[0].pars;
    if (p.globalUpdateInterval > 0 && performedSweeps_ % p.globalUpdateInterval == 0) {
        if (p.globalShift) attemptGlobalMove(g, MoveShift);
        if (p.wolffClusterUpdate) attemptGlobalMove(g, MoveWolff);
        if (p.wolffClusterShiftUpdate) attemptGlobalMove(g, MoveWolffShift);
    }
}

void DetSDW::syncPhiFromDevice(int b) {
    dqmc_ctx* ctx_ = select(b);
    check(dqmc_get_fields_host(ctx_, ch_[b].phi.data(), nullptr, nullptr), "dqmc_get_fields_host");
}

// detsdwopdim.cpp:4242-4300
double DetSDW::phiAction(const Chain& ch) const {
    const detsdw_params& pars = ch.pars;
    const double dtau = pars.dtau, r = pars.r, u = pars.u, c = pars.c;
    const int L = pars.L;
    const std::vector<double>& f = ch.phi;
    double action = 0.0;
    for (int k = 1; k <= m_; ++k) {
        const int kprev = (k > 1) ? k - 1 : m_;
        for (int site = 0; site < N_; ++site) {
            const int x = site % L, y = site / L;
            const int xn = y * L + (x + 1) % L, yn = ((y + 1) % L) * L + x;
            double phisq = 0.0;
            if (!pars.phi2bosons) {
                double td2 = 0.0, xd2 = 0.0, yd2 = 0.0;
                for (int d = 0; d < opdim_; ++d) {
                    const double ph = f[phiIdx(site, d, k)];
                    const double td = (ph - f[phiIdx(site, d, kprev)]) / dtau;
                    td2 += td * td;
                    const double xd = ph - f[phiIdx(xn, d, k)];
                    xd2 += xd * xd;
                    const double yd = ph - f[phiIdx(yn, d, k)];
                    yd2 += yd * yd;
                }
                action += (dtau / (2.0 * c * c)) * td2;
                action += 0.5 * dtau * xd2;
                action += 0.5 * dtau * yd2;
            }
            for (int d = 0; d < opdim_; ++d) phisq += f[phiIdx(site, d, k)] * f[phiIdx(site, d, k)];
            action += 0.5 * dtau * r * phisq;
            if (!pars.phi2bosons) action += 0.25 * dtau * u * phisq * phisq;
        }
    }
    return action;
}

// addGlobalRandomDisplacement (:3755-3763): all slices (incl. the unused slice 0) shifted
void DetSDW::addGlobalRandomDisplacement(Chain& c) {
    for (int dim = 0; dim < opdim_; ++dim) {
        const double rr = c.rng.randRange(-c.phiDelta, +c.phiDelta);
        for (int k = 0; k <= m_; ++k)
            for (int site = 0; site < N_; ++site) c.phi[phiIdx(site, dim, k)] += rr;
    }
}

// arma::dot on the OPDIM-vectors (op_dot::direct_dot_arma: two accumulators over even / odd elements)
static inline double adot(const double* a, const double* b, int n) {
    double v1 = 0.0, v2 = 0.0;
    int i = 0;
    for (; i + 1 < n; i += 2) { v1 += a[i] * b[i]; v2 += a[i + 1] * b[i + 1]; }
    if (i < n) v1 += a[i] * b[i];
    return v1 + v2;
}

// buildAndFlipCluster with randomDirection<OPDIM> (detsdwopdim.cpp:3765-3883): Wolff single cluster over the
// (site, time slice) lattice, reflection phi -> phi - 2 (phi . rd) rd; works on the host mirror of the field
unsigned DetSDW::buildAndFlipCluster(Chain& c) {
    const int L = c.pars.L;
    const double dtau = c.pars.dtau;
    double rd[3] = {0.0, 0.0, 0.0};
    if (opdim_ == 1) rd[0] = (c.rng.rand01() <= 0.5) ? -1.0 : +1.0;
    else if (opdim_ == 2) c.rng.randPointOnCircle(rd[0], rd[1]);
    else c.rng.randPointOnSphere(rd[0], rd[1], rd[2]);
    auto getPhi = [&](int site, int k, double* out) { for (int d = 0; d < opdim_; ++d) out[d] = c.phi[phiIdx(site, d, k)]; };
    auto projected = [&](int site, int k) { double ph[3]; getPhi(site, k, ph); return adot(ph, rd, opdim_); };
    auto flip = [&](int site, int k) {
        double ph[3];
        getPhi(site, k, ph);
        const double f = 2. * adot(ph, rd, opdim_);
        for (int d = 0; d < opdim_; ++d) c.phi[phiIdx(site, d, k)] = ph[d] - f * rd[d];
    };
    std::vector<char> visited((size_t)N_ * (m_ + 1), 0);
    std::vector<std::pair<int, int>> next_sites;                      // std::stack in the reference
    int timeslice = c.rng.randInt(1, m_);
    int site = c.rng.randInt(0, N_ - 1);
    flip(site, timeslice);
    visited[(size_t)timeslice * N_ + site] = 1;
    next_sites.push_back({site, timeslice});
    unsigned cluster_size = 1;
    do {
        site = next_sites.back().first; timeslice = next_sites.back().second;
        next_sites.pop_back();
        const int x = site % L, y = site / L;
        const int nb[4] = {y * L + (x + 1) % L, y * L + (x - 1 + L) % L, ((y + 1) % L) * L + x, ((y - 1 + L) % L) * L + x};
        for (int d = 0; d < 4; ++d) {                                 // XPLUS, XMINUS, YPLUS, YMINUS
            const int ns = nb[d];
            if (!visited[(size_t)timeslice * N_ + ns]) {
                const double bond_arg = 2. * dtau * projected(site, timeslice) * projected(ns, timeslice);
                if (bond_arg < 0 && c.rng.rand01() <= (1. - std::exp(bond_arg))) {
                    flip(ns, timeslice);
                    visited[(size_t)timeslice * N_ + ns] = 1;
                    next_sites.push_back({ns, timeslice});
                    ++cluster_size;
                }
            }
        }
        const int tn[2] = {timeslice < m_ ? timeslice + 1 : 1, timeslice > 1 ? timeslice - 1 : m_};   // ChainDir PLUS, MINUS
        for (int t = 0; t < 2; ++t) {
            const int nt = tn[t];
            if (!visited[(size_t)nt * N_ + site]) {
                const double bond_arg = (2. / dtau) * projected(site, timeslice) * projected(site, nt);
                if (bond_arg < 0 && c.rng.rand01() <= (1. - std::exp(bond_arg))) {
                    flip(site, nt);
                    visited[(size_t)nt * N_ + site] = 1;
                    next_sites.push_back({site, nt});
                    ++cluster_size;
                }
            }
        }
    } while (!next_sites.empty());
    return cluster_size;
}

// attemptGlobalShiftMove (detsdwopdim.cpp:3565-3644), attemptWolffClusterUpdate (:3488-3562),
// attemptWolffClusterShiftUpdate (:3647-3751) for every chain of the batch: the proposal of each chain is drawn from
// its own RNG stream on the host mirror of its field, the UdV storage / G of all chains are rebuilt by ONE batched
// set-up, then each chain accepts or restores on its own.
void DetSDW::attemptGlobalMove(Group& g, GlobalMoveKind kind) {
    const int nb = g.count;
    dqmc_ctx* ctx = g.ctx;
    const size_t nphi = (size_t)N_ * opdim_ * (m_ + 1);
    std::vector<double> prob_scalar(nb, 1.0), old_sv((size_t)nb * ng_), new_sv((size_t)nb * ng_), added(nb, 0.0);
    std::vector<dqmc_update_state> st(nb);
    check(dqmc_get_update_states_all_host(ctx, st.data()), "dqmc_get_update_states_all_host");
    check(dqmc_get_sv_all_host(ctx, old_sv.data()), "dqmc_get_sv_all_host");
    // The pure shift move never leaves the device: both bosonic actions are reductions over the resident field and the
    // displacement is a constant per component; the host only draws the displacement (the chain's RNG stream, reference order).
    // Rounding: k_phi_action adds 256 strided partial sums and then a tree, the reference (phiAction, :4242-4300) adds slice by slice
    // and site by site -- the two actions (~1e4) agree to ~1e-12 absolute, so exp(-(s_new - s_old)) agrees to ~1e-12 relative and a
    // decision differs from the reference's only if its uniform falls within that distance of the probability (once in ~1e12 moves;
    // the same holds for the fermionic ratio, a product of n_g singular-value quotients).  The host mirror c.phi is NOT touched here:
    // it is valid only after syncPhiFromDevice (every reader calls it).
    const bool on_device = (kind == MoveShift);
    if (on_device) {
        std::vector<double> s_old(nb), s_new(nb), shifts((size_t)nb * opdim_);
        check(dqmc_phi_action_all_host(ctx, s_old.data()), "phiAction");
        check(dqmc_backup(ctx), "globalMoveStoreBackups");
        for (int b = 0; b < nb; ++b) {
            Chain& c = ch_[g.first + b];
            c.phiDelta = st[b].phiDelta;
            for (int dim = 0; dim < opdim_; ++dim) shifts[(size_t)b * opdim_ + dim] = c.rng.randRange(-c.phiDelta, +c.phiDelta);   // :3755-3763
        }
        check(dqmc_shift_fields_all_host(ctx, shifts.data()), "addGlobalRandomDisplacement");
        check(dqmc_phi_action_all_host(ctx, s_new.data()), "phiAction");
        for (int b = 0; b < nb; ++b) prob_scalar[b] = std::exp(-(s_new[b] - s_old[b]));
    } else {
    // ONE transfer each for the fields of all chains of the group
    g.fields.resize(nphi * nb);
    check(dqmc_get_fields_all_host(ctx, g.fields.data()), "dqmc_get_fields_all_host");
    check(dqmc_backup(ctx), "globalMoveStoreBackups");
    for (int b = 0; b < nb; ++b) {
        Chain& c = ch_[g.first + b];
        std::memcpy(c.phi.data(), &g.fields[nphi * b], nphi * sizeof(double));     // g.fields keeps the backup copy
        c.phiDelta = st[b].phiDelta;
        if (kind != MoveShift)
            for (int r = 0; r < c.pars.repeatWolffPerSweep; ++r) added[b] += (double)buildAndFlipCluster(c);
        if (kind != MoveWolff) {
            const double old_scalar_action = phiAction(c);            // for the combined move: after the cluster flips
            addGlobalRandomDisplacement(c);
            const double new_scalar_action = phiAction(c);
            prob_scalar[b] = std::exp(-(new_scalar_action - old_scalar_action));
        }
    }
    {
        std::vector<double> proposed(nphi * nb);
        for (int b = 0; b < nb; ++b) std::memcpy(&proposed[nphi * b], ch_[g.first + b].phi.data(), nphi * sizeof(double));
        check(dqmc_set_fields_all_host(ctx, proposed.data()), "updateCoshSinhTermsPhi");
    }
    }
    setupUdVStorage_and_calculateGreen(g);
    check(dqmc_get_sv_all_host(ctx, new_sv.data()), "dqmc_get_sv_all_host");
    for (int b = 0; b < nb; ++b) {
        Chain& c = ch_[g.first + b];
        double log_prob = 0.0;
        for (int j = 0; j < ng_; ++j) log_prob += std::log(new_sv[(size_t)b * ng_ + j]) - std::log(old_sv[(size_t)b * ng_ + j]);
        double prob_fermion = std::exp(log_prob);
        if (opdim_ < 3) prob_fermion = prob_fermion * prob_fermion;
        const double prob = (kind == MoveWolff) ? prob_fermion : prob_scalar[b] * prob_fermion;
        int& attempted = kind == MoveShift ? c.attemptedGlobalShifts : kind == MoveWolff ? c.attemptedWolffClusterUpdates
                                                                                          : c.attemptedWolffClusterShiftUpdates;
        int& accepted = kind == MoveShift ? c.acceptedGlobalShifts : kind == MoveWolff ? c.acceptedWolffClusterUpdates
                                                                                        : c.acceptedWolffClusterShiftUpdates;
        attempted += 1;
        if (prob >= 1.0 || c.rng.rand01() < prob) {
            accepted += 1;
            c.addedWolffClusterSize += added[b];
        } else {
            check(dqmc_select_chain(ctx, b), "dqmc_select_chain");
            check(dqmc_restore(ctx), "globalMoveRestoreBackups");
            if (!on_device) std::memcpy(c.phi.data(), &g.fields[nphi * b], nphi * sizeof(double));
        }
    }
}

void DetSDW::set_exchange_parameter_value(double r, int b) {
    dqmc_ctx* ctx_ = select(b);
    ch_[b].pars.r = r;
    check(dqmc_set_exchange_parameter(ctx_, r), "set_exchange_parameter_value");
}
double DetSDW::get_exchange_action_contribution(int b) {
    double v = 0.0;
    dqmc_ctx* ctx_ = select(b);
    check(dqmc_exchange_action_host(ctx_, &v), "get_exchange_action_contribution");
    return v;
}
void DetSDW::exchangeActionsDevice(double* out_dev) {
    for (auto& g : groups_) check(dqmc_exchange_actions_device(g.ctx, out_dev + g.first), "get_exchange_action_contribution (device)");
}
void DetSDW::get_control_data(detsdw_control_data& out, int b) {
    dqmc_ctx* ctx_ = select(b);
    out.acceptedGlobalShifts = ch_[b].acceptedGlobalShifts;
    out.attemptedGlobalShifts = ch_[b].attemptedGlobalShifts;
    out.acceptedWolffClusterUpdates = ch_[b].acceptedWolffClusterUpdates;
    out.attemptedWolffClusterUpdates = ch_[b].attemptedWolffClusterUpdates;
    out.acceptedWolffClusterShiftUpdates = ch_[b].acceptedWolffClusterShiftUpdates;
    out.attemptedWolffClusterShiftUpdates = ch_[b].attemptedWolffClusterShiftUpdates;
    out.addedWolffClusterSize = ch_[b].addedWolffClusterSize;
    check(dqmc_get_update_state_host(ctx_, &out.adjust), "get_control_data");
}
void DetSDW::set_control_data(const detsdw_control_data& in, int b) {
    dqmc_ctx* ctx_ = select(b);
    ch_[b].acceptedGlobalShifts = in.acceptedGlobalShifts;
    ch_[b].attemptedGlobalShifts = in.attemptedGlobalShifts;
    ch_[b].acceptedWolffClusterUpdates = in.acceptedWolffClusterUpdates;
    ch_[b].attemptedWolffClusterUpdates = in.attemptedWolffClusterUpdates;
    ch_[b].acceptedWolffClusterShiftUpdates = in.acceptedWolffClusterShiftUpdates;
    ch_[b].attemptedWolffClusterShiftUpdates = in.attemptedWolffClusterShiftUpdates;
    ch_[b].addedWolffClusterSize = in.addedWolffClusterSize;
    dqmc_update_state st = in.adjust;
    st.rng_consumed = 0; st.rng_avail = 0; st.error = 0;      // the RNG window is per replica, never exchanged
    check(dqmc_set_update_state_host(ctx_, &st), "set_control_data");
    ch_[b].phiDelta = st.phiDelta;
    ch_[b].lastAccRatio = st.lastAccRatio;
    ch_[b].angleDelta = st.angleDelta;
    ch_[b].scaleDelta = st.scaleDelta;
}

void DetSDW::getInfo(detsdw_info& o, int b) {
    dqmc_ctx* ctx_ = select(b);
    const Chain& c = ch_[b];
    std::memset(&o, 0, sizeof(o));
    o.opdim = opdim_; o.L = c.pars.L; o.N = N_; o.MSF = MSF_; o.n_g = ng_; o.m = m_; o.s = s_; o.n = n_;
    o.performedSweeps = performedSweeps_; o.lastSweepDir = (int)lastSweepDir_;
    o.acceptedGlobalShifts = c.acceptedGlobalShifts; o.attemptedGlobalShifts = c.attemptedGlobalShifts;
    o.acceptedWolffClusterUpdates = c.acceptedWolffClusterUpdates; o.attemptedWolffClusterUpdates = c.attemptedWolffClusterUpdates;
    o.acceptedWolffClusterShiftUpdates = c.acceptedWolffClusterShiftUpdates;
    o.attemptedWolffClusterShiftUpdates = c.attemptedWolffClusterShiftUpdates;
    o.addedWolffClusterSize = c.addedWolffClusterSize;
    o.currentTimeslice = dqmc_current_timeslice(ctx_);
    o.beta = c.pars.beta; o.dtau = c.pars.dtau; o.phiDelta = c.phiDelta; o.lastAccRatioLocal_phi = c.lastAccRatio;
    o.r = c.pars.r; o.rngDrawn = c.rng.drawn();
    o.angleDelta = c.angleDelta; o.scaleDelta = c.scaleDelta;
}
void DetSDW::getPhi(double* out, int b) {
    syncPhiFromDevice(b);
    std::memcpy(out, ch_[b].phi.data(), ch_[b].phi.size() * sizeof(double));
}
// cdwl(site, k) as the reference's MatInt (N x (m+1), column-major); all +-1 / +-2, slice 0 unused
void DetSDW::getCdwl(int32_t* out, int b) {
    if (ch_[b].pars.cdwU != 0.0) check(dqmc_get_cdwl_host(select(b), ch_[b].cdwl.data()), "dqmc_get_cdwl_host");
    std::memcpy(out, ch_[b].cdwl.data(), ch_[b].cdwl.size() * sizeof(int32_t));
}
void DetSDW::setCdwl(const int32_t* in, int b) {
    if (ch_[b].pars.cdwU == 0.0) throw ParameterWrong("setCdwl: the replica was created with cdwU == 0");
    dqmc_ctx* ctx_ = select(b);
    std::memcpy(ch_[b].cdwl.data(), in, ch_[b].cdwl.size() * sizeof(int32_t));
    check(dqmc_set_cdwl_host(ctx_, ch_[b].cdwl.data()), "dqmc_set_cdwl_host");
    setupUdVStorage_and_calculateGreen(grp(b));
    lastSweepDir_ = Up;
}
// also rebuilds UdV storage and G -- of every chain of the batch (one batched setup)
void DetSDW::setPhi(const double* in, int b) {
    dqmc_ctx* ctx_ = select(b);
    std::memcpy(ch_[b].phi.data(), in, ch_[b].phi.size() * sizeof(double));
    check(dqmc_set_fields_host(ctx_, ch_[b].phi.data()), "dqmc_set_fields_host");
    setupUdVStorage_and_calculateGreen(grp(b));             // one batched set-up: every chain of b's sub-batch is rebuilt
    lastSweepDir_ = Up;
}
// detsdwopdim.cpp:4991-5012
void DetSDW::saveConfigurationStreamBinary(const std::string& directory, int b) {
    syncPhiFromDevice(b);
    const std::string path = directory + "/configs-phi.binarystream";
    std::FILE* f = std::fopen(path.c_str(), "ab");
    if (!f) throw GeneralError(DQMC_EINVAL, "Could not open file " + path + " for writing");
    const int L = ch_[b].pars.L;
    bool ok = true;
    for (int ix = 0; ix < L; ++ix)
        for (int iy = 0; iy < L; ++iy) {
            const int i = iy * L + ix;
            for (int k = 1; k <= m_; ++k)
                for (int dim = 0; dim < opdim_; ++dim) ok &= std::fwrite(&ch_[b].phi[phiIdx(i, dim, k)], sizeof(double), 1, f) == 1;
        }
    ok &= std::fclose(f) == 0;
    if (!ok) throw GeneralError(DQMC_EINVAL, "write error on " + path);
    if (ch_[b].pars.cdwU != 0.0) {           // configs-l.binarystream (:5015-5037): the discrete field as int32, same site order
        check(dqmc_get_cdwl_host(select(b), ch_[b].cdwl.data()), "dqmc_get_cdwl_host");
        const std::string lpath = directory + "/configs-l.binarystream";
        std::FILE* fl = std::fopen(lpath.c_str(), "ab");
        if (!fl) throw GeneralError(DQMC_EINVAL, "Could not open file " + lpath + " for writing");
        bool okl = true;
        for (int ix = 0; ix < L; ++ix)
            for (int iy = 0; iy < L; ++iy)
                for (int k = 1; k <= m_; ++k) okl &= std::fwrite(&ch_[b].cdwl[(size_t)k * N_ + iy * L + ix], sizeof(int32_t), 1, fl) == 1;
        okl &= std::fclose(fl) == 0;
        if (!okl) throw GeneralError(DQMC_EINVAL, "write error on " + lpath);
    }
}
// ---- checkpoint / resume --------------------------------------------------------------------------------------
namespace {
const char kMagic[8] = {'D', 'Q', 'M', 'C', 'K', 'P', 'T', '1'};
struct FileCloser { std::FILE* f; ~FileCloser() { if (f) std::fclose(f); } };
void wr(std::FILE* f, const void* p, size_t n) { if (std::fwrite(p, 1, n, f) != n) throw GeneralError(DQMC_EINVAL, "checkpoint: write error"); }
void rd(std::FILE* f, void* p, size_t n) { if (std::fread(p, 1, n, f) != n) throw GeneralError(DQMC_EINVAL, "checkpoint: truncated file"); }
}  // namespace

void DetSDW::saveState(const std::string& path) {
    FileCloser fc{std::fopen(path.c_str(), "wb")};
    if (!fc.f) throw GeneralError(DQMC_EINVAL, "Could not open file " + path + " for writing");
    const int32_t hdr[4] = {(int32_t)ch_.size(), (int32_t)sizeof(detsdw_params), performedSweeps_, (int32_t)lastSweepDir_};
    wr(fc.f, kMagic, 8); wr(fc.f, hdr, sizeof(hdr));
    for (int b = 0; b < (int)ch_.size(); ++b) {
        Chain& c = ch_[b];
        syncPhiFromDevice(b);
        detsdw_control_data cd;
        get_control_data(cd, b);
        const std::vector<uint64_t> rng = c.rng.serialize();
        const uint64_t nr = rng.size(), np = c.phi.size();
        wr(fc.f, &c.pars, sizeof(c.pars)); wr(fc.f, &cd, sizeof(cd));
        wr(fc.f, &nr, 8); wr(fc.f, rng.data(), nr * 8);
        wr(fc.f, &np, 8); wr(fc.f, c.phi.data(), np * 8);
        if (c.pars.cdwU != 0.0) {            // the discrete field follows phi (records of a cdwU == 0 replica are unchanged)
            check(dqmc_get_cdwl_host(select(b), c.cdwl.data()), "dqmc_get_cdwl_host");
            wr(fc.f, c.cdwl.data(), c.cdwl.size() * sizeof(int32_t));
        }
    }
}

void DetSDW::loadState(const std::string& path) {
    FileCloser fc{std::fopen(path.c_str(), "rb")};
    if (!fc.f) throw GeneralError(DQMC_EINVAL, "Could not open file " + path + " for reading");
    char magic[8]; int32_t hdr[4];
    rd(fc.f, magic, 8); rd(fc.f, hdr, sizeof(hdr));
    if (std::memcmp(magic, kMagic, 8) != 0) throw GeneralError(DQMC_EINVAL, "checkpoint: not a detqmc_amd state file");
    if (hdr[0] != (int32_t)ch_.size() || hdr[1] != (int32_t)sizeof(detsdw_params))
        throw GeneralError(DQMC_EINVAL, "checkpoint: written for a different number of replicas / library version");
    for (int b = 0; b < (int)ch_.size(); ++b) {
        Chain& c = ch_[b];
        detsdw_params p; detsdw_control_data cd;
        rd(fc.f, &p, sizeof(p)); rd(fc.f, &cd, sizeof(cd));
        // r is restored from the file; the device may differ
        if (!same_model(p, c.pars, /*seeds_too=*/true)) throw GeneralError(DQMC_EINVAL, "checkpoint: parameters differ from this replica's");
        uint64_t nr = 0, np = 0;
        rd(fc.f, &nr, 8);
        if (nr < DSFMT19937::state_words() + 3 || nr > (1u << 26)) throw GeneralError(DQMC_EINVAL, "checkpoint: bad RNG record");
        std::vector<uint64_t> rng(nr);
        rd(fc.f, rng.data(), nr * 8);
        rd(fc.f, &np, 8);
        if (np != c.phi.size()) throw GeneralError(DQMC_EINVAL, "checkpoint: field size mismatch");
        rd(fc.f, c.phi.data(), np * 8);
        if (c.pars.cdwU != 0.0) rd(fc.f, c.cdwl.data(), c.cdwl.size() * sizeof(int32_t));
        c.rng.deserialize(rng);
        dqmc_ctx* ctx_ = select(b);
        c.pars.r = p.r;
        check(dqmc_set_exchange_parameter(ctx_, p.r), "dqmc_set_exchange_parameter");
        check(dqmc_set_fields_host(ctx_, c.phi.data()), "dqmc_set_fields_host");
        if (c.pars.cdwU != 0.0) check(dqmc_set_cdwl_host(ctx_, c.cdwl.data()), "dqmc_set_cdwl_host");
        set_control_data(cd, b);
    }
    performedSweeps_ = hdr[2];
    forEachGroup([this](Group& g) { setupUdVStorage_and_calculateGreen(g); });   // like the reference's resume: G(beta) from scratch, next sweep goes down
    lastSweepDir_ = Up;
}

void DetSDW::getGreen(dqmc_cplx* g, int b) { check(dqmc_get_green_host(select(b), g), "dqmc_get_green_host"); }
void DetSDW::getGreenInvSv(double* sv, int b) { check(dqmc_get_sv_host(select(b), sv), "dqmc_get_sv_host"); }

}  // namespace detqmc

// ---------------------------------------------------------------------------------------------
// C API (include/detsdw_host.h)
// ---------------------------------------------------------------------------------------------
using detqmc::DetSDW;
struct detsdw_replica { DetSDW* impl; int sel; };   // sel: chain the per-replica calls refer to
static thread_local std::string g_host_err;

#define RGUARD(...)                                                                          \
    if (!r) { g_host_err = "null replica handle"; return DQMC_EINVAL; }                      \
    GUARD(__VA_ARGS__)
#define GUARD(...)                                                          \
    try { __VA_ARGS__; return DQMC_OK; }                                            \
    catch (const detqmc::GeneralError& e) { g_host_err = e.what(); return e.code; } \
    catch (const std::exception& e) { g_host_err = e.what(); return DQMC_EINVAL; }

extern "C" const char* detsdw_last_error(void) { return g_host_err.c_str(); }

extern "C" int detsdw_create(const detsdw_params* p, detsdw_replica** out) {
    if (!p || !out) { g_host_err = "null argument"; return DQMC_EINVAL; }
    *out = nullptr;
    GUARD({ DetSDW* d = new DetSDW(*p); *out = new detsdw_replica{d, 0}; })
}
extern "C" int detsdw_create_batch(const detsdw_params* p, int nchains, detsdw_replica** out) {
    if (!p || !out) { g_host_err = "null argument"; return DQMC_EINVAL; }
    *out = nullptr;
    GUARD({ DetSDW* d = new DetSDW(p, nchains); *out = new detsdw_replica{d, 0}; })
}
extern "C" int detsdw_create_batch_ex(const detsdw_params* p, int nchains, int sub_batches, detsdw_replica** out) {
    if (!p || !out) { g_host_err = "null argument"; return DQMC_EINVAL; }
    *out = nullptr;
    GUARD({ DetSDW* d = new DetSDW(p, nchains, sub_batches); *out = new detsdw_replica{d, 0}; })
}
extern "C" int detsdw_num_sub_batches(detsdw_replica* r) { return r ? r->impl->numSubBatches() : 0; }
extern "C" dqmc_ctx* detsdw_ctx_of_chain(detsdw_replica* r, int chain, int* local_index) {
    if (!r || chain < 0 || chain >= r->impl->numChains()) return nullptr;
    return r->impl->ctx(chain, local_index);
}
extern "C" int detsdw_num_chains(detsdw_replica* r) { return r ? r->impl->numChains() : 0; }
extern "C" int detsdw_select_chain(detsdw_replica* r, int chain) {
    if (!r || chain < 0 || chain >= r->impl->numChains()) { g_host_err = "chain index out of range"; return DQMC_EINVAL; }
    r->sel = chain;
    return DQMC_OK;
}
extern "C" void detsdw_destroy(detsdw_replica* r) { if (r) { delete r->impl; delete r; } }
extern "C" int detsdw_sweep(detsdw_replica* r, int tm) { RGUARD(r->impl->sweep(tm != 0)) }
extern "C" int detsdw_sweep_thermalization(detsdw_replica* r) { RGUARD(r->impl->sweepThermalization()) }
extern "C" int detsdw_get_info(detsdw_replica* r, detsdw_info* out) { RGUARD(r->impl->getInfo(*out, r->sel)) }
extern "C" int detsdw_get_observables(detsdw_replica* r, detsdw_observables* out) {
    RGUARD(r->impl->getObservables(*out, r->sel))
}
extern "C" int detsdw_get_observable_vector(detsdw_replica* r, int which, double* out) {
    RGUARD(r->impl->getObservableVector(which, out, r->sel))
}
extern "C" int detsdw_get_phi(detsdw_replica* r, double* phi) { RGUARD(r->impl->getPhi(phi, r->sel)) }
extern "C" int detsdw_set_phi(detsdw_replica* r, const double* phi) { RGUARD(r->impl->setPhi(phi, r->sel)) }
extern "C" int detsdw_get_cdwl(detsdw_replica* r, int32_t* cdwl) { RGUARD(r->impl->getCdwl(cdwl, r->sel)) }
extern "C" int detsdw_set_cdwl(detsdw_replica* r, const int32_t* cdwl) { RGUARD(r->impl->setCdwl(cdwl, r->sel)) }
extern "C" int detsdw_get_green(detsdw_replica* r, dqmc_cplx* g) { RGUARD(r->impl->getGreen(g, r->sel)) }
extern "C" int detsdw_get_green_inv_sv(detsdw_replica* r, double* sv) { RGUARD(r->impl->getGreenInvSv(sv, r->sel)) }
extern "C" int detsdw_save_state(detsdw_replica* r, const char* path) {
    if (!path) { g_host_err = "null path"; return DQMC_EINVAL; }
    RGUARD(r->impl->saveState(path))
}
extern "C" int detsdw_load_state(detsdw_replica* r, const char* path) {
    if (!path) { g_host_err = "null path"; return DQMC_EINVAL; }
    RGUARD(r->impl->loadState(path))
}
extern "C" int detsdw_save_configuration_stream_binary(detsdw_replica* r, const char* directory) {
    RGUARD(r->impl->saveConfigurationStreamBinary(directory ? directory : ".", r->sel))
}
extern "C" double detsdw_rng_rand01(detsdw_replica* r) { return r ? r->impl->rand01(r->sel) : -1.0; }
extern "C" dqmc_ctx* detsdw_ctx(detsdw_replica* r) { return r ? r->impl->ctx(r->sel) : nullptr; }
extern "C" double detsdw_get_exchange_parameter_value(detsdw_replica* r) { return r ? r->impl->get_exchange_parameter_value(r->sel) : 0.0; }
extern "C" int detsdw_set_exchange_parameter_value(detsdw_replica* r, double v) { RGUARD(r->impl->set_exchange_parameter_value(v, r->sel)) }
extern "C" const char* detsdw_get_exchange_parameter_name(detsdw_replica* r) { return r ? r->impl->get_exchange_parameter_name() : ""; }
extern "C" int detsdw_get_exchange_action_contribution(detsdw_replica* r, double* out) {
    RGUARD(*out = r->impl->get_exchange_action_contribution(r->sel))
}
extern "C" int detsdw_exchange_actions_device(detsdw_replica* r, double* out_dev) {
    if (!out_dev) { g_host_err = "null argument"; return DQMC_EINVAL; }
    RGUARD(r->impl->exchangeActionsDevice(out_dev))
}
extern "C" int detsdw_get_control_data(detsdw_replica* r, detsdw_control_data* out) { RGUARD(r->impl->get_control_data(*out, r->sel)) }
extern "C" int detsdw_set_control_data(detsdw_replica* r, const detsdw_control_data* in) { RGUARD(r->impl->set_control_data(*in, r->sel)) }
// detsdwopdim.cpp:5251-5264 (Hukushima & Nemoto 1996)
extern "C" double detsdw_replica_exchange_probability(double par1, double action1, double par2, double action2) {
    const double delta = (par1 - par2) * (action2 - action1);
    return delta <= 0.0 ? 1.0 : std::exp(-delta);
}
extern "C" int detsdw_rng_fill(uint32_t seed, uint32_t processIndex, double* out, size_t n) {
    if (!out) return DQMC_EINVAL;
    detqmc::RngStream rs(seed, processIndex);
    for (size_t i = 0; i < n; ++i) out[i] = rs.rand01();
    return DQMC_OK;
}
