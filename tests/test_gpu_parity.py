"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against
  (1) the committed golden fixtures generated from the REAL reference (tests/golden/*.npz), and
  (2) the CPU oracle (oracle/detsdw_oracle.py) on the same seeded inputs.
Tolerance: 1e-10 relative for fp64 Green's functions / singular values (BASELINE.json north_star);
field configurations and the RNG stream position must agree exactly (same Markov chain).
Nothing here reads /root/reference.
"""
import numpy as np
import pytest

from conftest import load_golden, oracle_params, relerr, relerr_blocks

pytestmark = pytest.mark.gpu

TOL = 1e-10
SMALL = ["o2_L4", "o2_L4_s7", "o2_L4_flux", "o2_L4_apbc", "o1_L4", "o3_L4", "o2_L6_seed",
         "o2_L4_dense", "o2_L4_dense_flux"]     # *_dense: checkerboard=false (CB_NONE)
# round 4: flux / antiperiodic boundaries at L = 8 -- 16 plaquettes per subgroup, Landau-gauge phases for y = 0 .. 7, the
# boundary-crossing vertical phases, APBC sign flips, mu_x != mu_y, s not dividing m, CB_NONE + flux
# (/root/reference/src/detsdwopdim.cpp:1598-1684, 1788-1826)
L8FLUX = ["o2_L8_flux", "o2_L8_apbc_flux", "o2_L8_dense_flux"]
L8 = L8FLUX + ["o2_L8_apbc"]


def _ctx_from_params(a, **over):
    from detqmc_amd import KernelContext
    op = oracle_params(a).finalize()
    kw = dict(opdim=op.opdim, L=op.L, m=op.m, s=op.s, dtau=op.dtau, delaySteps=op.delaySteps, bc=op.bc,
              weakZflux=op.weakZflux, r=op.r, c=op.c, u=op.u, lambda_=op.lambda_, txhor=op.txhor,
              txver=op.txver, tyhor=op.tyhor, tyver=op.tyver, mux=op.mux, muy=op.muy, accRatio=op.accRatio,
              checkerboard=op.checkerboard, cdwU=op.cdwU)
    kw.update(over)
    return KernelContext(**kw), op


def _sdw_params(a, **over):
    from detqmc_amd import SDWParams
    op = oracle_params(a)
    kw = dict(opdim=op.opdim, L=op.L, beta=op.beta, dtau=op.dtau, s=op.s, r=op.r, c=op.c, u=op.u,
              lambda_=op.lambda_, txhor=op.txhor, txver=op.txver, tyhor=op.tyhor, tyver=op.tyver, mu=op.mu,
              mux=op.mux, muy=op.muy, accRatio=op.accRatio, delaySteps=op.delaySteps, bc=op.bc,
              weakZflux=op.weakZflux, globalShift=op.globalShift, globalUpdateInterval=op.globalUpdateInterval,
              wolffClusterUpdate=op.wolffClusterUpdate, wolffClusterShiftUpdate=op.wolffClusterShiftUpdate,
              repeatWolffPerSweep=op.repeatWolffPerSweep, fermionMeasurements=not op.turnoffFermionMeasurements,
              rngSeed=op.rngSeed, simindex=op.simindex, checkerboard=op.checkerboard, cdwU=op.cdwU,
              spinProposalMethod=op.spinProposalMethod, adaptScaleVariance=op.adaptScaleVariance, repeatUpdateInSlice=op.repeatUpdateInSlice)
    kw.update(over)
    return SDWParams(**kw)


def _golden_phi(g, key):
    return np.transpose(g[key], (2, 0, 1))        # (N, OPDIM, m+1) -> (m+1, N, OPDIM)


# ------------------------------------------------------------------------------------------------
# dense products on the matrix cores
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L", [4, 6, 8])
def test_gemm_all_ops(L):
    from detqmc_amd import KernelContext
    ctx = KernelContext(2, L, 20, 10, 0.1, delaySteps=4)
    n = ctx.ng
    rng = np.random.default_rng(7)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    for opA in (0, 1):
        for opB in (0, 1):
            ref = (A.conj().T if opA else A) @ (B.conj().T if opB else B)
            got = ctx.gemm(opA, opB, A, B)
            assert relerr(got, ref) < 1e-13, (opA, opB)
    # identity with an asymmetric partner catches transposed fragment layouts
    I = np.eye(n)
    assert relerr(ctx.gemm(0, 0, I, B), B) < 1e-15
    assert relerr(ctx.gemm(0, 0, A, I), A) < 1e-15
    ctx.close()


# ------------------------------------------------------------------------------------------------
# checkerboard B-multiplies (a10-a14)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5"] + L8FLUX)
def test_bmult_vs_reference(name):
    from detsdw_oracle import make_test_matrix
    g = load_golden(name)
    ctx, op = _ctx_from_params(g["params"])
    ctx.set_fields(_golden_phi(g, "init_phi"))
    phi, ch, sh = ctx.get_fields()
    assert np.array_equal(phi[1:], _golden_phi(g, "init_phi")[1:])
    if "init_coshTermPhi" in g:
        assert relerr(ch[1:], g["init_coshTermPhi"].T[1:]) < 1e-14
        assert relerr(sh[1:], g["init_sinhTermPhi"].T[1:]) < 1e-14
    A = make_test_matrix(ctx.ng)
    k = int(g["bmult_k"][0])
    assert relerr(ctx.leftMultiplyBmat(A, k, k - 1), g["bmult_left"]) < 1e-12
    assert relerr(ctx.rightMultiplyBmat(A, k, k - 1), g["bmult_right"]) < 1e-12
    if "bmult_leftinv" in g:
        assert relerr(ctx.leftMultiplyBmatInv(A, k, k - 1), g["bmult_leftinv"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmatInv(A, k, k - 1), g["bmult_rightinv"]) < 1e-12
    if "bchain_left" in g:
        k2 = int(g["bchain_k2"][0])
        assert relerr(ctx.leftMultiplyBmat(A, k2, 0), g["bchain_left"]) < 1e-12
        assert relerr(ctx.leftMultiplyBmatInv(A, k2, 0), g["bchain_leftinv"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmat(A, k2, 0), g["bchain_right"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmatInv(A, k2, 0), g["bchain_rightinv"]) < 1e-12
    if not op.checkerboard:       # CB_NONE: the multiply IS the dense computeBmatSDW (detsdwopdim.cpp:1309-1485)
        assert relerr(ctx.leftMultiplyBmat(np.eye(ctx.ng), k, k - 1), g["bdense_k"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmat(np.eye(ctx.ng), k, k - 1), g["bdense_k"]) < 1e-12
    # exact-to-rounding identity of the symmetric break-up: B^-1 B = 1
    # (error grows with the condition number of the chain, so keep it to one stabilisation interval)
    R = ctx.leftMultiplyBmatInv(ctx.leftMultiplyBmat(A, op.s, 0), op.s, 0)
    assert relerr(R, A) < 1e-10
    R = ctx.rightMultiplyBmatInv(ctx.rightMultiplyBmat(A, op.s, 0), op.s, 0)
    assert relerr(R, A) < 1e-10
    ctx.close()


# ------------------------------------------------------------------------------------------------
# UdV decomposition (a2)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,opdim", [(4, 2), (6, 2), (8, 2), (4, 3)])
def test_udv_decompose(L, opdim):
    from detqmc_amd import KernelContext
    ctx = KernelContext(opdim, L, 20, 10, 0.1, delaySteps=4)
    n = ctx.ng
    rng = np.random.default_rng(11)
    Q1, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Q2, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    for d_true in (np.linspace(2.0, 0.5, n), np.logspace(12, -12, n)):
        M = (Q1 * d_true[None, :]) @ Q2.conj().T
        U, d, Vt, sweeps = ctx.udvDecompose(M)
        assert np.all(np.diff(d) <= 0), "singular values must be sorted descending like zgesvd"
        sref = np.linalg.svd(M, compute_uv=False)
        assert np.max(np.abs(d - sref) / sref[0]) < 1e-13
        assert relerr(U.conj().T @ U, np.eye(n)) < 1e-12
        assert relerr(Vt.conj().T @ Vt, np.eye(n)) < 1e-12
        assert relerr((U * d[None, :]) @ Vt.conj().T, M) < 1e-12
        assert 1 <= sweeps <= 40
    # column-graded matrix (the UdV chain's shape): high RELATIVE accuracy of every singular value
    W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    dd = np.logspace(10, -10, n)
    M = W * dd[None, :]
    U, d, Vt, _ = ctx.udvDecompose(M)
    import scipy.linalg as sla
    # reference values from the same matrix with columns rescaled in extended precision is overkill;
    # one-sided Jacobi in LAPACK (zgesvj) has the same relative-accuracy property
    sref = sla.svd(M, compute_uv=False, lapack_driver="gesvd")
    assert np.max(np.abs(d - sref) / sref[0]) < 1e-13
    assert relerr((U * d[None, :]) @ Vt.conj().T, M) < 1e-12
    ctx.close()


# ------------------------------------------------------------------------------------------------
# stabilised Green's function from scratch and through advance / wrap (a3-a8)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5"] + L8)
def test_green_from_scratch_vs_reference(name):
    g = load_golden(name)
    ctx, op = _ctx_from_params(g["params"])
    ctx.set_fields(_golden_phi(g, "init_phi"))
    ctx.setupUdVStorage_and_calculateGreen()
    assert ctx.currentTimeslice == op.m
    assert relerr(ctx.g, g["init_g"]) < TOL
    assert relerr(ctx.g_inv_sv, g["init_g_inv_sv"]) < TOL
    for l in range(op.n + 1):
        U, d, Vt = ctx.udv(l)
        assert np.max(np.abs(d - g["init_udv_d"][:, l]) / g["init_udv_d"][:, l]) < 1e-9
        if f"init_udv_U_{l}" in g and l > 0:
            # U d V^H is unique even though U, V are only fixed up to phases
            ref = (g[f"init_udv_U_{l}"] * g["init_udv_d"][:, l][None, :]) @ g[f"init_udv_Vt_{l}"].conj().T
            assert relerr((U * d[None, :]) @ Vt.conj().T, ref) < 1e-10
    ctx.close()


@pytest.mark.parametrize("name", ["o2_L4", "o2_L4_s7", "o2_L6_seed", "o3_L4"])
def test_wrap_advance_without_updates_vs_oracle(name):
    """One down sweep and one up sweep with the fields frozen: every advance must reproduce the
    oracle's G; wrapped G must agree with the freshly advanced one (the reference's
    --logGreenConsistency self-check, detsdwopdim.cpp:4831-4856)."""
    from detsdw_oracle import DetSDWOracle
    g = load_golden(name)
    ctx, op = _ctx_from_params(g["params"])
    o = DetSDWOracle(oracle_params(g["params"]))
    ctx.set_fields(o.phi)
    ctx.setupUdVStorage_and_calculateGreen()
    n, s, m = op.n, op.s, op.m
    for k in range(m, (n - 1) * s, -1):
        ctx.wrapDownGreen(k); o.wrapDownGreen(k)
    assert relerr(ctx.g, o.g) < TOL
    for l in range(n - 1, 0, -1):
        wrapped = ctx.g
        ctx.advanceDownGreen(l + 1); o.advanceDownGreen(l + 1)
        assert relerr(ctx.g, o.g) < TOL
        assert relerr(ctx.g_inv_sv, o.g_inv_sv) < TOL
        assert relerr(wrapped, ctx.g) < 1e-7
        for k in range(l * s, (l - 1) * s, -1):
            ctx.wrapDownGreen(k); o.wrapDownGreen(k)
    ctx.advanceDownGreen(1); o.advanceDownGreen(1)
    assert relerr(ctx.g, o.g) < TOL
    # up
    ctx.reset_storage0()
    from detsdw_oracle import UdV
    o.UdVStorage[0] = UdV.eye(o.ng)
    for l in range(0, n - 1):
        for k in range(l * s + 1, (l + 1) * s + 1):
            ctx.wrapUpGreen(k - 1); o.wrapUpGreen(k - 1)
        ctx.advanceUpGreen(l); o.advanceUpGreen(l)
        assert relerr(ctx.g, o.g) < TOL
    for k in range((n - 1) * s + 1, m + 1):
        ctx.wrapUpGreen(k - 1); o.wrapUpGreen(k - 1)
    ctx.advanceUpGreen(n - 1); o.advanceUpGreen(n - 1)
    assert relerr(ctx.g, o.g) < TOL
    assert relerr(ctx.g_inv_sv, o.g_inv_sv) < TOL
    ctx.close()


# ------------------------------------------------------------------------------------------------
# local updates of one slice (a17-a20)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("threads", [256, 512])
@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5"] + L8)
def test_update_slice_vs_reference(name, threads):
    """threads: both launch shapes of the decision kernel (dqmc_tuning::decide_threads; O(3) has the 256-thread shape only)"""
    from dsfmt_oracle import RngWrapper
    g = load_golden(name)
    if threads == 512 and oracle_params(g["params"]).opdim == 3:
        pytest.skip("O(3): one launch shape")
    ctx, op = _ctx_from_params(g["params"], decideThreads=threads)
    phi0 = _golden_phi(g, "init_phi")
    ctx.set_fields(phi0)
    ctx.setupUdVStorage_and_calculateGreen()
    # the reference's stream: (OPDIM+1) draws per site and slice went into the random field
    r = RngWrapper(op.rngSeed, op.simindex + 1)
    for _ in range((op.opdim + 1) * op.N * op.m):
        r.rand01()
    window = np.array([r.rand01() for _ in range((op.opdim + 1) * op.N)])
    ctx.push_uniforms(window)
    ctx.updateInSlice(op.m, thermalization=True)
    st = ctx.update_state()
    phi, ch, sh = ctx.get_fields()
    assert np.array_equal(phi[op.m], g["slice_phi_m"]), "accept/reject decisions differ from the reference"
    assert relerr(ctx.g, g["slice_g"]) < TOL
    assert relerr_blocks(ctx.g, g["slice_g"]) < 1e-9       # every 16 x 16 block on its own scale
    assert abs(st.lastAccRatio - g["slice_accRatio"][0]) < 1e-15
    assert st.ra_samplesAdded == 1
    # cosh/sinh caches consistent with the new field (reference consistencyCheck, detsdwopdim.cpp:4617-4667)
    nrm = np.sqrt(np.sum(phi[op.m] ** 2, axis=1))
    assert relerr(ch[op.m], np.cosh(op.lambda_ * op.dtau * nrm)) < 1e-13
    assert relerr(sh[op.m], np.sinh(op.lambda_ * op.dtau * nrm) / nrm) < 1e-13
    ctx.wrapDownGreen(op.m)
    if "slice_g_wrapped" in g:
        assert relerr(ctx.g, g["slice_g_wrapped"]) < TOL
    ctx.close()


def test_update_slice_delay_steps_invariance():
    """The chain must not depend on the delay depth (reference: iterative / woodbury / delayed are
    alternative update methods for the same move, detsdwopdim.cpp:2493-3175)."""
    from dsfmt_oracle import RngWrapper
    g = load_golden("o2_L4")
    res = []
    for D in (1, 3, 6, 16):
        ctx, op = _ctx_from_params(g["params"], delaySteps=D)
        ctx.set_fields(_golden_phi(g, "init_phi"))
        ctx.setupUdVStorage_and_calculateGreen()
        r = RngWrapper(op.rngSeed, op.simindex + 1)
        for _ in range((op.opdim + 1) * op.N * op.m):
            r.rand01()
        ctx.push_uniforms(np.array([r.rand01() for _ in range((op.opdim + 1) * op.N)]))
        ctx.updateInSlice(op.m, thermalization=False)
        res.append((ctx.get_fields()[0][op.m], ctx.g, ctx.update_state().rng_consumed))
        ctx.close()
    for phi, G, used in res[1:]:
        assert np.array_equal(phi, res[0][0])
        assert relerr(G, res[0][1]) < 1e-11
        assert used == res[0][2]
    assert np.array_equal(res[0][0], g["slice_phi_m"])


def test_rng_window_exhaustion_is_reported():
    from detqmc_amd import DqmcError
    g = load_golden("o2_L4")
    ctx, op = _ctx_from_params(g["params"])
    ctx.set_fields(_golden_phi(g, "init_phi"))
    ctx.setupUdVStorage_and_calculateGreen()
    ctx.push_uniforms(np.full(5, 0.5))
    ctx.updateInSlice(op.m)
    with pytest.raises(DqmcError) as e:
        ctx.update_state()
    assert e.value.code == -4
    ctx.close()


# ------------------------------------------------------------------------------------------------
# whole sweeps through the C++ host layer (a9) -- identical Markov chain as the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5", "o2_L4_gshift"] + L8)
def test_replica_trajectory_vs_reference(name):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"]))
    info = rep.info
    if "init_phi" in g:
        assert np.array_equal(rep.phi[1:], _golden_phi(g, "init_phi")[1:])
        assert relerr(rep.g, g["init_g"]) < TOL
        assert relerr(rep.g_inv_sv, g["init_g_inv_sv"]) < TOL
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: trajectory diverged"
        assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL, f"sweep {i}"
        assert relerr_blocks(rep.g, g[f"sweep{i}_g"]) < 1e-9, f"sweep {i}"
        assert relerr(rep.g_inv_sv, g[f"sweep{i}_g_inv_sv"]) < TOL
        inf = rep.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert abs(inf.lastAccRatioLocal_phi - g[f"sweep{i}_lastAccRatio"][0]) < 1e-15
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    assert i > 2
    act = rep.get_exchange_action_contribution()
    assert abs(act - g["exchange_action"][0]) < 1e-12 * abs(g["exchange_action"][0])
    nxt = np.array([rep.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"]), "RNG stream position differs from the reference"
    assert info.n_g == g["init_g"].shape[0]
    rep.close()


# ------------------------------------------------------------------------------------------------
# cdwU != 0: the discrete field l_i(tau) in e^{dtau V} and its own update pass
# ------------------------------------------------------------------------------------------------
def _golden_cdwl(g, key):
    return np.ascontiguousarray(g[key].T).astype(np.int32)          # (N, m+1) -> (m+1, N)


@pytest.mark.parametrize("name", ["o2_L4_cdw_slice", "o3_L4_cdw_slice", "o1_L4_cdw", "o2_L4_cdw_dense", "o2_L4_cdw_gshift"])
def test_cdw_bmult_green_and_one_slice_vs_reference(name):
    """The reference's B-multiplies, G(beta) and (where the fixture has it) ONE updateInSlice -- phi pass, then the cdwl pass -- at
    cdwU != 0, through the kernel-level ABI: fields set from the fixture, uniforms from the reference's stream."""
    from detsdw_oracle import make_test_matrix
    from dsfmt_oracle import RngWrapper
    g = load_golden(name)
    ctx, op = _ctx_from_params(g["params"])
    assert op.cdwU != 0
    ctx.set_fields(_golden_phi(g, "init_phi"))
    l0 = _golden_cdwl(g, "init_cdwl")
    l0[0] = 1                                                        # slice 0 is unused (0 in the reference's matrix)
    ctx.set_cdwl(l0)
    assert np.array_equal(ctx.get_cdwl()[1:], l0[1:])
    A = make_test_matrix(ctx.ng)
    if "bmult_left" in g:
        k = int(g["bmult_k"][0])
        assert relerr(ctx.leftMultiplyBmat(A, k, k - 1), g["bmult_left"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmat(A, k, k - 1), g["bmult_right"]) < 1e-12
        assert relerr(ctx.leftMultiplyBmatInv(A, k, k - 1), g["bmult_leftinv"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmatInv(A, k, k - 1), g["bmult_rightinv"]) < 1e-12
        k2 = int(g["bchain_k2"][0])
        assert relerr(ctx.leftMultiplyBmat(A, k2, 0), g["bchain_left"]) < 1e-12
        assert relerr(ctx.rightMultiplyBmatInv(A, k2, 0), g["bchain_rightinv"]) < 1e-12
    ctx.setupUdVStorage_and_calculateGreen()
    assert relerr(ctx.g, g["init_g"]) < TOL
    assert relerr(ctx.g_inv_sv, g["init_g_inv_sv"]) < TOL
    if "slice_cdwl_m" in g:
        r = RngWrapper(op.rngSeed, op.simindex + 1)
        for _ in range((op.opdim + 1) * op.N * op.m):
            r.rand01()
        ctx.push_uniforms(np.array([r.rand01() for _ in range((op.opdim + 3) * op.N)]))
        ctx.updateInSlice(op.m, thermalization=True)
        st = ctx.update_state()
        assert np.array_equal(ctx.get_fields()[0][op.m], g["slice_phi_m"]), "phi pass: accept/reject decisions differ from the reference"
        assert np.array_equal(ctx.get_cdwl()[op.m], g["slice_cdwl_m"].reshape(-1).astype(np.int32)), "cdwl pass differs from the reference"
        assert relerr(ctx.g, g["slice_g"]) < TOL
        assert abs(st.lastAccRatio - g["slice_accRatio"][0]) < 1e-15, "the cdwl pass must not touch the phi acceptance ratio"
        assert st.ra_samplesAdded == 1
        ctx.wrapDownGreen(op.m)
        assert relerr(ctx.g, g["slice_g_wrapped"]) < TOL
    ctx.close()


@pytest.mark.parametrize("name", ["o2_L4_cdw", "o1_L4_cdw", "o2_L4_cdw_gshift", "o2_L4_cdw_dense"])
def test_cdw_replica_trajectory_vs_reference(name):
    """Whole sweeps at cdwU != 0 through the host layer: phi, the discrete field, G and the RNG position after every sweep are the
    reference's (seeds: oracle/find_cdw_seeds.py -- the reference's last-bit branch at null cdwl proposals)."""
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"]))
    assert np.array_equal(rep.phi[1:], _golden_phi(g, "init_phi")[1:])
    assert np.array_equal(rep.cdwl[1:], _golden_cdwl(g, "init_cdwl")[1:])
    assert relerr(rep.g, g["init_g"]) < TOL
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: phi trajectory diverged"
        assert np.array_equal(rep.cdwl[1:], _golden_cdwl(g, f"sweep{i}_cdwl")[1:]), f"sweep {i}: cdwl trajectory diverged"
        assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL, f"sweep {i}"
        inf = rep.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert abs(inf.lastAccRatioLocal_phi - g[f"sweep{i}_lastAccRatio"][0]) < 1e-15
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    assert i > 1
    assert np.array_equal(np.array([rep.rand01() for _ in range(4)]), g["rng_next"]), "RNG stream position differs from the reference"
    rep.close()


@pytest.mark.parametrize("opdim,L,D", [(3, 4, 6), (2, 6, 8), (3, 6, 12)])
def test_cdw_trajectory_vs_oracle(opdim, L, D):
    """Sizes and O(3) trajectories the reference cannot pin (its null-proposal branch is a coin flip there): the HIP path against the
    oracle, same seed => same chain, two sweeps, in a batch of two replicas with different r."""
    from detsdw_oracle import DetSDWOracle, SDWParams as OP
    from detqmc_amd import DetSDWBatch, SDWParams
    common = dict(opdim=opdim, L=L, beta=1.2, s=5, delaySteps=D, cdwU=0.6, rngSeed=4711)
    ps = [SDWParams(**common, r=-1.0, simindex=0), SDWParams(**common, r=-0.4, simindex=1)]
    batch = DetSDWBatch(ps)
    os_ = [DetSDWOracle(OP(**common, r=-1.0, simindex=0)), DetSDWOracle(OP(**common, r=-0.4, simindex=1))]
    for sweep in range(2):
        batch.sweepThermalization()
        for b, o in enumerate(os_):
            o.sweepThermalization()
            rep = batch.chain(b)
            assert np.array_equal(rep.phi[1:], o.phi[1:]), (sweep, b)
            assert np.array_equal(rep.cdwl[1:], o.cdwl[1:]), (sweep, b)
            assert relerr(rep.g, o.g) < TOL
    for b, o in enumerate(os_):
        assert np.array_equal(np.array([batch.chain(b).rand01() for _ in range(3)]), np.array([o.rng.rand01() for _ in range(3)]))
    batch.close()


def test_cdw_headline_size_properties():
    """cdwU != 0 at the headline lattice (L = 16, n_g = 512, delay depth 32, QR stabilisation, a batch of two replicas), checked through
    properties that do not need the oracle at this size: B^-1 B = 1 with the discrete field in e^{-+dtau V}, the wrapped Green's function
    against the one rebuilt from the fields after two sweeps, the field l stays in {+-1, +-2} and does change, the caches of phi stay exact,
    and a second handle with the same seeds in a different delay depth walks the same chain."""
    from detqmc_amd import DetSDWBatch, SDWParams
    from detsdw_oracle import make_test_matrix
    common = dict(opdim=2, L=16, beta=2.0, s=10, cdwU=0.45, stabilisation="qr", rngSeed=20262026)
    out = []
    for D in (32, 16):
        batch = DetSDWBatch([SDWParams(**common, delaySteps=D, r=-1.0, simindex=0), SDWParams(**common, delaySteps=D, r=-0.6, simindex=1)])
        l0 = batch.chain(1).cdwl
        for _ in range(2):
            batch.sweepThermalization()
        rep = batch.chain(1)
        l1, phi = rep.cdwl, rep.phi
        assert set(np.unique(l1[1:])) <= {-2, -1, 1, 2} and np.any(l1[1:] != l0[1:])
        ctx = rep.kernel_context
        if D == 32:
            A = make_test_matrix(512)
            assert relerr(ctx.leftMultiplyBmatInv(ctx.leftMultiplyBmat(A, 10, 0), 10, 0), A) < 1e-10
            assert relerr(ctx.rightMultiplyBmatInv(ctx.rightMultiplyBmat(A, 10, 0), 10, 0), A) < 1e-10
            _, ch, sh = ctx.get_fields()
            nrm = np.sqrt(np.sum(phi[1:] ** 2, axis=2))
            assert relerr(ch[1:], np.cosh(0.1 * nrm)) < 1e-14 and relerr(sh[1:], np.sinh(0.1 * nrm) / nrm) < 1e-13
        G = rep.g
        out.append((phi, l1, G, rep.info.rngDrawn))
        # G carried through two sweeps of wraps, updates and re-stabilisations against G rebuilt from the fields
        ctx.setupUdVStorage_and_calculateGreen()
        assert relerr(G, ctx.g) < 1e-8
        batch.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][3] == out[1][3]
    assert relerr(out[0][2], out[1][2]) < 1e-9


def test_cdw_checkpoint_and_config_stream(tmp_path):
    """The discrete field travels with the checkpoint (saved after an even number of sweeps, like the phi-only test further down) and
    is written as configs-l.binarystream next to configs-phi.binarystream (detsdwopdim.cpp:5015-5037)."""
    from detqmc_amd import DetSDW
    g = load_golden("o2_L4_cdw")
    rep = DetSDW(_sdw_params(g["params"]))
    for _ in range(2):
        rep.sweepThermalization()
    path = str(tmp_path / "state.bin")
    rep.save_state(path)
    rep.saveConfigurationStreamBinary(str(tmp_path))
    l1 = rep.cdwl
    assert np.array_equal(l1[1:], _golden_cdwl(g, "sweep2_cdwl")[1:])
    raw = np.fromfile(str(tmp_path / "configs-l.binarystream"), dtype=np.int32)
    info = rep.info
    Lx = info.L
    want = np.array([l1[k, iy * Lx + ix] for ix in range(Lx) for iy in range(Lx) for k in range(1, info.m + 1)], dtype=np.int32)
    assert np.array_equal(raw, want)
    for _ in range(2):
        rep.sweepThermalization()
    phi2, l2, drawn = rep.phi, rep.cdwl, rep.info.rngDrawn
    rep.close()
    rep = DetSDW(_sdw_params(g["params"]))
    rep.load_state(path)
    assert np.array_equal(rep.cdwl[1:], l1[1:])
    for _ in range(2):
        rep.sweepThermalization()
    assert np.array_equal(rep.phi[1:], phi2[1:]) and np.array_equal(rep.cdwl[1:], l2[1:]) and rep.info.rngDrawn == drawn
    rep.close()


def test_measurement_sweeps_do_not_adapt_step_size():
    from detqmc_amd import DetSDW
    g = load_golden("o2_L4")
    rep = DetSDW(_sdw_params(g["params"]))
    for _ in range(2):
        rep.sweep(False)
    assert rep.info.phiDelta == 0.5
    assert rep.get_control_data().adjust.ra_samplesAdded == 0
    rep.close()


def test_update_methods_give_the_same_chain():
    from detqmc_amd import DetSDW
    g = load_golden("o2_L4")
    out = []
    for um in ("delayed", "woodbury", "iterative"):
        rep = DetSDW(_sdw_params(g["params"], updateMethod=um))
        rep.sweepThermalization()
        rep.sweepThermalization()
        out.append((rep.phi, rep.g))
        rep.close()
    for phi, G in out[1:]:
        assert np.array_equal(phi, out[0][0])
        assert relerr(G, out[0][1]) < TOL
    assert np.array_equal(out[0][0][1:], _golden_phi(g, "sweep2_phi")[1:])


# ------------------------------------------------------------------------------------------------
# headline size (BASELINE config 3): checksums from the reference + size-independent properties
# ------------------------------------------------------------------------------------------------
def test_headline_size_vs_reference_checksums():
    from detqmc_amd import DetSDW
    from detsdw_oracle import make_test_matrix
    g = load_golden("o2_L16_b10")
    rep = DetSDW(_sdw_params(g["params"]))
    G = rep.g
    assert relerr(G[::16, ::16], g["init_g_sub16"]) < TOL
    assert relerr(np.diag(G), g["init_g_diag"]) < TOL
    assert abs(np.linalg.norm(G) - g["init_g_fro"][0]) < TOL * g["init_g_fro"][0]
    assert relerr(rep.g_inv_sv, g["init_g_inv_sv"]) < TOL
    for i in (1, 2):
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        G = rep.g
        assert relerr(G[::16, ::16], g[f"sweep{i}_g_sub16"]) < TOL
        assert relerr(np.diag(G), g[f"sweep{i}_g_diag"]) < TOL
        assert relerr(rep.g_inv_sv, g[f"sweep{i}_g_inv_sv"]) < TOL
    # properties that need no reference data
    ctx = rep.kernel_context
    A = make_test_matrix(ctx.ng)
    assert relerr(ctx.leftMultiplyBmatInv(ctx.leftMultiplyBmat(A, 10, 0), 10, 0), A) < 1e-10
    assert relerr(ctx.rightMultiplyBmat(ctx.rightMultiplyBmatInv(A, 100, 90), 100, 90), A) < 1e-10
    U, d, Vt = ctx.udv(5)
    I = np.eye(ctx.ng)
    assert relerr(U.conj().T @ U, I) < 1e-11 and relerr(Vt.conj().T @ Vt, I) < 1e-11
    assert np.all(np.diff(d) <= 0)
    rep.close()


# ------------------------------------------------------------------------------------------------
# QR ("UDT") stabilisation mode: different factorisation, same Green's functions and same Markov chain
# ------------------------------------------------------------------------------------------------
# n_g up to 1024: all register-resident panel depths; 1296 and 2304 (O(3) L = 24, BASELINE config 5): 512-thread panels;
# 2704: 1024-thread panels
@pytest.mark.parametrize("L,opdim", [(4, 2), (6, 2), (8, 2), (4, 3), (16, 2), (12, 3), (16, 3), (18, 3), (24, 3), (26, 3)])
def test_qr_udt_decompose(L, opdim):
    from detqmc_amd import KernelContext
    ctx = KernelContext(opdim, L, 20, 10, 0.1, delaySteps=4, stabilisation="qr")
    n = ctx.ng
    rng = np.random.default_rng(5)
    W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    for M in (W, W * np.logspace(8, -8, n)[None, :]):
        U, d, Vt, _ = ctx.udvDecompose(M)
        assert relerr(U.conj().T @ U, np.eye(n)) < 1e-12, "Q must be unitary"
        assert np.all(d > 0)
        assert relerr((U * d[None, :]) @ Vt.conj().T, M) < 1e-12
        if n > 1300:
            continue                                  # the LAPACK cross-checks below take too long on the host
        # T = V_t^H is a row-scaled, column-permuted triangular factor: well conditioned
        assert np.linalg.cond(Vt) < 1e6
        # |det M| is carried by d alone
        sref = np.linalg.svd(M, compute_uv=False)
        assert abs(np.sum(np.log(d)) + np.log(abs(np.linalg.det(Vt.conj().T))) - np.sum(np.log(sref))) < 1e-8 * n
    ctx.close()


@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5"] + L8)
def test_qr_mode_green_from_scratch_vs_reference(name):
    g = load_golden(name)
    ctx, op = _ctx_from_params(g["params"], stabilisation="qr")
    ctx.set_fields(_golden_phi(g, "init_phi"))
    ctx.setupUdVStorage_and_calculateGreen()
    assert relerr(ctx.g, g["init_g"]) < TOL
    # log|det G^-1| agrees with the sum over the reference's singular values
    assert abs(np.sum(np.log(ctx.g_inv_sv)) - np.sum(np.log(g["init_g_inv_sv"]))) < 1e-9 * ctx.ng
    ctx.close()


@pytest.mark.parametrize("name", SMALL + ["o2_L8_b5", "o2_L4_gshift"] + L8)
def test_qr_mode_replica_trajectory_vs_reference(name):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr"))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: trajectory diverged"
        assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL, f"sweep {i}"
        assert relerr_blocks(rep.g, g[f"sweep{i}_g"]) < 1e-9, f"sweep {i}"
        assert abs(np.sum(np.log(rep.g_inv_sv)) - np.sum(np.log(g[f"sweep{i}_g_inv_sv"]))) < 1e-9 * rep.info.n_g
        inf = rep.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    nxt = np.array([rep.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])
    rep.close()


@pytest.mark.parametrize("threads", [256, 512])
@pytest.mark.parametrize("delaySteps", [16, 32])
def test_qr_mode_headline_size_vs_reference_checksums(delaySteps, threads):
    """delaySteps = 32 (bench.py's setting) against the fixture the reference produced with 16: the depth of the
    delayed-update blocks is a performance knob, the Markov chain does not depend on it -- nor on the launch shape of the
    decision kernel (256 threads: what a context of 128 chains runs; 512: what a single chain runs)."""
    from detqmc_amd import DetSDW
    g = load_golden("o2_L16_b10")
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", delaySteps=delaySteps, decideThreads=threads))
    G = rep.g
    assert relerr(G[::16, ::16], g["init_g_sub16"]) < TOL
    assert relerr(np.diag(G), g["init_g_diag"]) < TOL
    assert abs(np.sum(np.log(rep.g_inv_sv)) - np.sum(np.log(g["init_g_inv_sv"]))) < 1e-9 * 512
    for i in (1, 2):
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        G = rep.g
        assert relerr(G[::16, ::16], g[f"sweep{i}_g_sub16"]) < TOL
        assert relerr(np.diag(G), g[f"sweep{i}_g_diag"]) < TOL
        assert abs(np.linalg.norm(G) - g[f"sweep{i}_g_fro"][0]) < TOL * g[f"sweep{i}_g_fro"][0]
    rep.close()


# ------------------------------------------------------------------------------------------------
# colder (beta = 20, BASELINE config 4's temperature) and larger O(3) cases, both stabilisation modes
# ------------------------------------------------------------------------------------------------
def _check_subsampled(G, g, tag, ss):
    assert relerr(G[::ss, ::ss], g[f"{tag}_g_sub{ss}"]) < TOL
    assert relerr(np.diag(G), g[f"{tag}_g_diag"]) < TOL
    assert abs(np.linalg.norm(G) - g[f"{tag}_g_fro"][0]) < TOL * g[f"{tag}_g_fro"][0]


@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", ["o2_L8_b20", "o3_L6"])
def test_cold_and_large_o3_trajectories_vs_reference(name, stab):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation=stab))
    _check_subsampled(rep.g, g, "init", 4)
    assert abs(np.sum(np.log(rep.g_inv_sv)) - np.sum(np.log(g["init_g_inv_sv"]))) < 1e-9 * rep.info.n_g
    for i in (1, 2):
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        _check_subsampled(rep.g, g, f"sweep{i}", 4)
        assert abs(np.sum(np.log(rep.g_inv_sv)) - np.sum(np.log(g[f"sweep{i}_g_inv_sv"]))) < 1e-9 * rep.info.n_g
    nxt = np.array([rep.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])
    rep.close()


# ------------------------------------------------------------------------------------------------
# batched chains: nb replicas in lockstep through one context (grid.z = chain)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", ["o2_L4_gshift", "o3_L4", "o2_L6_seed", "o2_L4_wolff", "o3_L4_wolff"])
def test_batched_chains_follow_the_single_replica_chains(name, stab):
    """Chain 0 of the batch has the fixture's parameters -> must reproduce the REFERENCE trajectory; the other
    chains (other seeds, other r) must be identical to single-replica runs with the same parameters."""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch
    g = load_golden(name)
    p0 = _sdw_params(g["params"], stabilisation=stab)
    plist = [p0,
             dataclasses.replace(p0, r=p0.r + 0.4, rngSeed=777, simindex=1),
             dataclasses.replace(p0, r=p0.r - 0.3, rngSeed=p0.rngSeed, simindex=5)]
    batch = DetSDWBatch(plist)
    singles = [DetSDW(p) for p in plist[1:]]
    nsweeps = 0
    while f"sweep{nsweeps + 1}_phi" in g:
        nsweeps += 1
    for i in range(1, nsweeps + 1):
        batch.sweepThermalization()
        for sgl in singles:
            sgl.sweepThermalization()
        c0 = batch.chain(0)
        assert np.array_equal(c0.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: chain 0 left the reference chain"
        assert relerr(c0.g, g[f"sweep{i}_g"]) < TOL
        inf = c0.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        for b, sgl in enumerate(singles, start=1):
            cb = batch.chain(b)
            assert np.array_equal(cb.phi, sgl.phi), f"sweep {i}: chain {b} differs from its single-replica run"
            assert relerr(cb.g, sgl.g) < 1e-12
            ib, isg = cb.info, sgl.info
            assert ib.phiDelta == isg.phiDelta and ib.rngDrawn == isg.rngDrawn
            assert ib.acceptedGlobalShifts == isg.acceptedGlobalShifts
            assert abs(cb.get_exchange_action_contribution() - sgl.get_exchange_action_contribution()) < 1e-12
    nxt = np.array([batch.chain(0).rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])
    for sgl in singles:
        sgl.close()
    batch.close()


def test_batch_rejects_replicas_that_differ_in_more_than_r_and_seed():
    import dataclasses
    from detqmc_amd import DetSDWBatch, DqmcError
    g = load_golden("o2_L4")
    p0 = _sdw_params(g["params"])
    with pytest.raises(DqmcError):
        DetSDWBatch([p0, dataclasses.replace(p0, u=p0.u + 0.1)])


def test_batched_kernel_context_entry_points():
    """gemm / bmult / decompose through a 3-chain context: results of the selected chain; per-chain fields."""
    from detqmc_amd import KernelContext
    from detsdw_oracle import DetSDWOracle, make_test_matrix
    g = load_golden("o2_L4")
    ctx, op = _ctx_from_params(g["params"], nchains=3)
    o = DetSDWOracle(op)
    phi = _golden_phi(g, "init_phi")
    A = make_test_matrix(ctx.ng)
    k = 3
    rng = np.random.default_rng(3)
    fields = [phi, phi * 0.5, phi + 0.1 * rng.standard_normal(phi.shape)]
    for b in range(3):
        ctx.select_chain(b)
        ctx.set_fields(fields[b])
    for b in range(3):
        ctx.select_chain(b)
        o.phi = fields[b].copy()
        o.updateCoshSinhTermsPhi()
        assert relerr(ctx.leftMultiplyBmat(A, k, 0), o.leftMultiplyBmat(A, k, 0)) < 1e-12, b
        assert relerr(ctx.rightMultiplyBmatInv(A, k, 0), o.rightMultiplyBmatInv(A, k, 0)) < 1e-12, b
        B = rng.standard_normal(A.shape) + 1j * rng.standard_normal(A.shape)
        assert relerr(ctx.gemm(0, 1, A, B), A @ B.conj().T) < 1e-13
    ctx.close()


def test_eight_chain_batch_matches_single_replicas():
    """8 chains take the XCD-aware launch shape of the wide MFMA kernels (chain -> XCD); same chains as single runs."""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch
    g = load_golden("o2_L6_seed")
    p0 = _sdw_params(g["params"], stabilisation="qr")
    plist = [dataclasses.replace(p0, simindex=p0.simindex + b, r=p0.r + 0.05 * b) for b in range(8)]
    batch = DetSDWBatch(plist)
    singles = [DetSDW(p) for p in plist]
    for i in range(2):
        batch.sweepThermalization()
        for sgl in singles:
            sgl.sweepThermalization()
    assert np.array_equal(batch.chain(0).phi[1:], _golden_phi(g, "sweep2_phi")[1:])
    for b, sgl in enumerate(singles):
        cb = batch.chain(b)
        assert np.array_equal(cb.phi, sgl.phi), b
        assert relerr(cb.g, sgl.g) < 1e-12, b
    for sgl in singles:
        sgl.close()
    batch.close()


def test_replica_exchange_between_the_chains_of_a_batch():
    """Parallel tempering inside one GPU: 4 batched chains on an r ladder exchange parameters (detqmc_amd/pt.py,
    no process group) exactly like 4 single replicas driven the same way."""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch
    from detqmc_amd.pt import ExchangeState, ReplicaAdapter, replica_exchange_step, replica_exchange_consistency_check
    g = load_golden("o2_L4")
    p0 = _sdw_params(g["params"], stabilisation="qr")
    rvals = [-1.2, -1.1, -1.0, -0.9]
    plist = [dataclasses.replace(p0, r=rvals[p], simindex=p) for p in range(4)]
    batch = DetSDWBatch(plist)
    breps = [ReplicaAdapter(batch.chain(b)) for b in range(4)]
    singles = [DetSDW(p) for p in plist]
    sreps = [ReplicaAdapter(s) for s in singles]
    stb = ExchangeState.create(rvals, 0, 1, n_local=4)
    sts = ExchangeState.create(rvals, 0, 1, n_local=4)
    moved = False
    for it in range(4):
        batch.sweepThermalization()
        for s in singles:
            s.sweepThermalization()
        ib = replica_exchange_step(breps, stb, None)
        isg = replica_exchange_step(sreps, sts, None)
        replica_exchange_consistency_check(breps, stb, None)
        assert ib == isg
        moved |= ib != [0, 1, 2, 3]
        for b in range(4):
            assert batch.chain(b).get_exchange_parameter_value() == rvals[ib[b]]
            assert np.array_equal(batch.chain(b).phi, singles[b].phi)
            assert batch.chain(b).info.phiDelta == singles[b].info.phiDelta
    assert moved, "ladder too steep for any swap: the test would not exercise the exchange of r on the device"
    for s in singles:
        s.close()
    batch.close()


# ------------------------------------------------------------------------------------------------
# Wolff cluster moves (a21): attemptWolffClusterUpdate / attemptWolffClusterShiftUpdate / buildAndFlipCluster
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", ["o2_L4_wolff", "o2_L4_wolffshift", "o1_L4_wolff", "o3_L4_wolff"])
def test_wolff_cluster_moves_vs_reference(name, stab):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation=stab))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: trajectory diverged"
        assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL, f"sweep {i}"
        inf = rep.info
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        assert inf.attemptedWolffClusterUpdates == int(g[f"sweep{i}_attWolff"][0])
        assert inf.acceptedWolffClusterUpdates == int(g[f"sweep{i}_accWolff"][0])
        assert inf.attemptedWolffClusterShiftUpdates == int(g[f"sweep{i}_attWolffShift"][0])
        assert inf.acceptedWolffClusterShiftUpdates == int(g[f"sweep{i}_accWolffShift"][0])
        assert inf.addedWolffClusterSize == g[f"sweep{i}_addedWolffClusterSize"][0]
        i += 1
    nxt = np.array([rep.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"]), "RNG stream position differs from the reference"
    rep.close()


def test_wolff_parameter_rules():
    """detsdwparams.cpp:89-96"""
    import dataclasses
    from detqmc_amd import DetSDW, DqmcError
    g = load_golden("o2_L4")
    p0 = _sdw_params(g["params"])
    with pytest.raises(DqmcError):
        DetSDW(dataclasses.replace(p0, wolffClusterShiftUpdate=True, globalShift=True))
    with pytest.raises(DqmcError):
        DetSDW(dataclasses.replace(p0, wolffClusterUpdate=True, globalUpdateInterval=0))


def test_configuration_stream_file_matches_the_reference(tmp_path):
    """saveConfigurationStreamBinary (detsdwopdim.cpp:4991-5012): byte-identical configs-phi.binarystream, appended twice
    like the reference harness did after the last sweep of the fixture."""
    from detqmc_amd import DetSDW
    g = load_golden("o2_L4_s7")
    rep = DetSDW(_sdw_params(g["params"]))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        i += 1
    rep.saveConfigurationStreamBinary(tmp_path)
    rep.saveConfigurationStreamBinary(tmp_path)
    got = np.fromfile(tmp_path / "configs-phi.binarystream", dtype=np.uint8)
    assert got.size == g["cfgstream_phi_bytes"].size == 2 * 16 * rep.info.m * 2 * 8
    assert np.array_equal(got, g["cfgstream_phi_bytes"])
    rep.close()


# ------------------------------------------------------------------------------------------------
# measurement sweeps: sweep(takeMeasurements=True), bosonic observables (reference with turnoffFermionMeasurements)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", ["o2_L4_meas", "o3_L4_meas"])
def test_measurement_sweeps_vs_reference(name, stab):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation=stab))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        i += 1
    assert not rep.observables.valid
    j = 1
    while f"meas{j}_phi" in g:
        rep.sweep(True)
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"meas{j}_phi")[1:]), f"measurement sweep {j}: trajectory diverged"
        assert relerr(rep.g, g[f"meas{j}_g"]) < TOL
        assert rep.info.phiDelta == g[f"meas{j}_phiDelta"][0], "measurement sweeps must not adapt the step size"
        o = rep.observables
        assert o.valid
        d = rep.info.opdim
        assert np.array_equal(np.array(o.meanPhi[:d]), g[f"meas{j}_meanPhi"].ravel())
        assert o.normMeanPhi == g[f"meas{j}_normMeanPhi"][0]
        assert o.associatedEnergy == g[f"meas{j}_associatedEnergy"][0]
        if d == 2:
            assert o.phiRhoS_Gc == g[f"meas{j}_phiRhoS_Gc"][0]
            assert o.phiRhoS_Gs == g[f"meas{j}_phiRhoS_Gs"][0]
        j += 1
    assert j > 2
    rep.close()


# ------------------------------------------------------------------------------------------------
# fermionic measurements on the device: shiftGreenSymmetric + G-dependent observables
# ------------------------------------------------------------------------------------------------
FMEAS = ["o2_L4_fmeas", "o2_L4_fmeas_apbc_flux", "o3_L4_fmeas", "o1_L4_fmeas", "o2_L8_fmeas_apbc_flux"]


@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", FMEAS)
def test_fermionic_measurement_sweeps_vs_reference(name, stab):
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation=stab))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        i += 1
    j = 1
    while f"meas{j}_phi" in g:
        rep.sweep(True)
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"meas{j}_phi")[1:]), f"measurement sweep {j}: trajectory diverged"
        o = rep.observables
        assert o.valid and o.fermionic_valid
        assert o.normMeanPhi == g[f"meas{j}_normMeanPhi"][0]
        for key in ("greenK0", "greenLocal", "pairPlusMax", "pairMinusMax", "occDiffSq"):
            want = g[f"meas{j}_{key}"][0]
            assert abs(getattr(o, key) - want) < TOL * max(1.0, abs(want)), (key, getattr(o, key), want)
        for key in ("kOccX", "kOccY", "pairPlus", "pairMinus"):
            assert relerr(rep.observable_vector(key), g[f"meas{j}_{key}"].ravel()) < TOL, key
        j += 1
    assert j > 1
    # shiftGreenSymmetric of the final state
    assert relerr(rep.kernel_context.shiftGreenSymmetric(), g["final_shiftGreenSymmetric"]) < TOL
    rep.close()


def test_fermionic_measurements_in_a_batch():
    """every chain of a batch measures like its single-replica twin"""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch
    g = load_golden("o2_L4_fmeas")
    p0 = _sdw_params(g["params"], stabilisation="qr")
    plist = [p0, dataclasses.replace(p0, simindex=3, r=-0.7)]
    batch = DetSDWBatch(plist)
    single = DetSDW(plist[1])
    for _ in range(2):
        batch.sweep(True)
        single.sweep(True)
    o0 = batch.chain(0).observables
    assert abs(o0.greenK0 - g["meas2_greenK0"][0]) < 1  # chain 0 started measuring one sweep earlier than the fixture: sanity only
    ob, os_ = batch.chain(1).observables, single.observables
    for key in ("greenK0", "greenLocal", "pairPlusMax", "pairMinusMax", "occDiffSq", "normMeanPhi"):
        assert getattr(ob, key) == getattr(os_, key), key
    for key in ("kOccX", "kOccY", "pairPlus", "pairMinus"):
        assert np.array_equal(batch.chain(1).observable_vector(key), single.observable_vector(key))
    single.close()
    batch.close()


# ------------------------------------------------------------------------------------------------
# checkpoint / resume
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("batched", [False, True])
def test_checkpoint_resume_continues_the_same_chain(tmp_path, batched):
    """state saved after an even number of sweeps; a fresh replica that loads it walks exactly the chain of the
    uninterrupted run (fields, step size, update statistics, RNG position), here with global moves every 2 sweeps"""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch, DqmcError
    g = load_golden("o2_L4_wolff")
    p0 = _sdw_params(g["params"], stabilisation="qr", globalUpdateInterval=2)
    plist = [p0, dataclasses.replace(p0, simindex=4, r=-0.8)] if batched else [p0]
    make = (lambda: DetSDWBatch(plist)) if batched else (lambda: DetSDW(p0))
    chains = (lambda rep: rep.chains) if batched else (lambda rep: [rep])
    a = make()
    for _ in range(4):
        a.sweepThermalization()
    ck = tmp_path / "state.ckpt"
    a.save_state(ck)
    for _ in range(4):
        a.sweepThermalization()
    b = make()
    b.load_state(ck)
    assert chains(b)[0].info.performedSweeps == 4
    for _ in range(4):
        b.sweepThermalization()
    for ca, cb in zip(chains(a), chains(b)):
        assert np.array_equal(ca.phi, cb.phi)
        ia, ib = ca.info, cb.info
        assert ia.phiDelta == ib.phiDelta and ia.rngDrawn == ib.rngDrawn
        assert ia.acceptedGlobalShifts == ib.acceptedGlobalShifts and ia.acceptedWolffClusterUpdates == ib.acceptedWolffClusterUpdates
        assert ia.addedWolffClusterSize == ib.addedWolffClusterSize
        assert relerr(cb.g, ca.g) < TOL
        assert ca.rand01() == cb.rand01()
    # a replica with other parameters refuses the file
    other = DetSDW(dataclasses.replace(p0, u=p0.u + 0.5))
    with pytest.raises(DqmcError):
        other.load_state(ck)
    other.close()
    a.close()
    b.close()


def test_headline_size_long_trajectory_batched_vs_reference():
    """6 sweeps at L = 16, beta = 10 with global shift moves every other sweep, run the way bench.py runs (batched
    replicas, QR stabilisation, delaySteps 32): chain 0 must reproduce the reference's field after EVERY sweep
    (SHA-256 of the raw cube), its Green's function checksums and its global-move statistics."""
    import dataclasses
    import hashlib
    from detqmc_amd import DetSDWBatch
    g = load_golden("o2_L16_b10_long")
    p0 = _sdw_params(g["params"], stabilisation="qr", delaySteps=32)
    batch = DetSDWBatch([p0, dataclasses.replace(p0, simindex=11)])
    c0 = batch.chain(0)

    def sha(phi_kNd):
        cube = np.asfortranarray(np.transpose(phi_kNd, (1, 2, 0)))          # (N, OPDIM, m+1) column-major, as dumped
        return np.frombuffer(hashlib.sha256(cube.tobytes(order="F")).digest(), dtype=np.uint8)

    assert g["init_phi_sha256"].size == 32
    i = 1
    while f"sweep{i}_phi_sha256" in g:
        batch.sweepThermalization()
        # the whole cube incl. the unused slice 0, which the global shift moves displace like every other slice
        assert np.array_equal(sha(c0.phi), g[f"sweep{i}_phi_sha256"]), f"sweep {i}: field differs from the reference"
        G = c0.g
        assert relerr(G[::32, ::32], g[f"sweep{i}_g_sub32"]) < TOL
        assert relerr(np.diag(G), g[f"sweep{i}_g_diag"]) < TOL
        inf = c0.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    assert i == 7
    nxt = np.array([c0.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])
    batch.close()


def test_o3_deepest_delay_blocks_vs_reference():
    """O(3) with MSF * delaySteps = 64 (the largest W the decision kernel holds: > 64 KiB of dynamic LDS, raised per
    instantiation) against the fixture the reference produced with delaySteps = 12"""
    from detqmc_amd import DetSDW
    g = load_golden("o3_L6")
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", delaySteps=16))
    for i in (1, 2):
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        _check_subsampled(rep.g, g, f"sweep{i}", 4)
    rep.close()


# ------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at FULL size, and one complete Green's function at the headline size
# ------------------------------------------------------------------------------------------------
def _sha_phi(phi_kNd):
    import hashlib
    cube = np.asfortranarray(np.transpose(phi_kNd, (1, 2, 0)))          # (N, OPDIM, m+1) column-major, as the harness dumps it
    return np.frombuffer(hashlib.sha256(cube.tobytes(order="F")).digest(), dtype=np.uint8)


def test_headline_size_full_green_function_vs_reference():
    """All 512 x 512 entries of G after the first sweep at L = 16, beta = 10 (the other headline fixtures keep
    sub-sampled checksums): a localised error off the sampled lattice cannot hide."""
    from detqmc_amd import DetSDW
    g = load_golden("o2_L16_b10_fullG")
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", delaySteps=32))
    assert np.array_equal(_sha_phi(rep.phi), g["init_phi_sha256"])
    rep.sweepThermalization()
    assert np.array_equal(_sha_phi(rep.phi), g["sweep1_phi_sha256"]), "field differs from the reference after sweep 1"
    G = rep.g
    assert G.shape == g["sweep1_g"].shape == (512, 512)
    assert relerr(G, g["sweep1_g"]) < TOL
    # entry-wise, not only in the max norm: every 16 x 16 block to 1e-9 of the block's own scale
    D = np.abs(G - g["sweep1_g"]).reshape(32, 16, 32, 16).max(axis=(1, 3))
    S = np.abs(g["sweep1_g"]).reshape(32, 16, 32, 16).max(axis=(1, 3))
    assert np.all(D <= 1e-9 * np.maximum(S, 1e-6))
    rep.close()


def test_config4_L16_beta20_eight_replicas_with_exchange_vs_reference():
    """BASELINE config 4 on one GPU: O(2) L = 16, beta = 20 (m = 200, n = 20), 8 replicas on an r ladder swept in
    lockstep by one context, global shift moves, replica exchange (detqmc_amd/pt.py, src/detqmcpt.h:963-1118).
    Sweeps 1-2: chain 0 (the fixture's parameters) must reproduce the reference's field (SHA-256), Green's function
    checksums, step size and global-move counters after every sweep.  Then exchange steps after every sweep: all 8
    chains must stay on the trajectories of 8 single-replica twins driven through the same exchange protocol."""
    import dataclasses
    from detqmc_amd import DetSDW, DetSDWBatch
    from detqmc_amd.pt import ExchangeState, ReplicaAdapter, replica_exchange_step, replica_exchange_consistency_check
    g = load_golden("o2_L16_b20")
    p0 = _sdw_params(g["params"], stabilisation="qr", delaySteps=32)
    assert p0.beta == 20 and p0.globalShift
    rvals = [p0.r + 0.004 * i for i in range(8)]
    plist = [dataclasses.replace(p0, r=rvals[i], simindex=i) for i in range(8)]
    batch = DetSDWBatch(plist)
    twins = [DetSDW(p) for p in plist]
    c0 = batch.chain(0)
    assert c0.info.m == 200 and c0.info.n == 20 and c0.info.n_g == 512
    assert np.array_equal(_sha_phi(c0.phi), g["init_phi_sha256"])
    G = c0.g
    assert relerr(G[::32, ::32], g["init_g_sub32"]) < TOL and relerr(np.diag(G), g["init_g_diag"]) < TOL
    assert abs(np.sum(np.log(c0.g_inv_sv)) - np.sum(np.log(g["init_g_inv_sv"]))) < 1e-9 * 512

    def step():
        batch.sweepThermalization()
        for t in twins:
            t.sweepThermalization()

    def compare_twins(tag):
        for b, t in enumerate(twins):
            cb = batch.chain(b)
            assert np.array_equal(cb.phi, t.phi), (tag, b)
            assert relerr(cb.g, t.g) < 1e-9, (tag, b)
            assert cb.info.phiDelta == t.info.phiDelta and cb.info.acceptedGlobalShifts == t.info.acceptedGlobalShifts, (tag, b)

    for i in (1, 2):
        step()
        assert np.array_equal(_sha_phi(c0.phi), g[f"sweep{i}_phi_sha256"]), f"sweep {i}: field differs from the reference"
        G = c0.g
        assert relerr(G[::32, ::32], g[f"sweep{i}_g_sub32"]) < TOL
        assert relerr(np.diag(G), g[f"sweep{i}_g_diag"]) < TOL
        assert abs(np.linalg.norm(G) - g[f"sweep{i}_g_fro"][0]) < TOL * g[f"sweep{i}_g_fro"][0]
        inf = c0.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert inf.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert inf.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
    compare_twins("before exchange")
    breps = [ReplicaAdapter(batch.chain(b)) for b in range(8)]
    treps = [ReplicaAdapter(t) for t in twins]
    stb = ExchangeState.create(rvals, 0, 1, n_local=8)
    stt = ExchangeState.create(rvals, 0, 1, n_local=8)
    for it in range(2):
        ib = replica_exchange_step(breps, stb, None)
        itw = replica_exchange_step(treps, stt, None)
        replica_exchange_consistency_check(breps, stb, None)
        assert ib == itw and sorted(ib) == list(range(8))
        for b in range(8):
            assert batch.chain(b).get_exchange_parameter_value() == rvals[ib[b]]
        step()
        compare_twins(f"after exchange {it}")
    assert stb.par_swapUpProposed == [2] * 7 + [0]
    for t in twins:
        t.close()
    batch.close()


def test_config5_O3_L24_beta20_full_size():
    """BASELINE config 5's size (O(3) L = 24, beta = 20: n_g = 2304, m = 200, n = 20; without the flux, which the
    reference rejects for opdim = 3, src/detsdwparams.cpp:57-60).  G(beta) and log det against the reference's
    construction (20 SVDs of 2304 x 2304 on the CPU), ONE slice of local updates and the wrap behind it against the reference,
    then a thermalisation sweep with size-independent properties: B^-1 B = 1, wrapped vs re-stabilised G, cosh/sinh caches,
    unitarity of the chain factor."""
    from detqmc_amd import DetSDW
    g = load_golden("o3_L24_b20_init")
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr"))
    inf = rep.info
    assert (inf.n_g, inf.m, inf.n) == (2304, 200, 20)
    assert np.array_equal(_sha_phi(rep.phi), g["init_phi_sha256"])
    G = rep.g
    assert relerr(G[::64, ::64], g["init_g_sub64"]) < TOL
    assert relerr(np.diag(G), g["init_g_diag"]) < TOL
    assert abs(np.linalg.norm(G) - g["init_g_fro"][0]) < TOL * g["init_g_fro"][0]
    assert abs(np.sum(np.log(rep.g_inv_sv)) - np.sum(np.log(g["init_g_inv_sv"]))) < 1e-9 * 2304
    ctx = rep.kernel_context
    # wrapped vs re-stabilised Green's function (the reference's greenConsistencyCheck, src/detmodel.h:960-1010):
    # wrap G(beta) down through the top interval without updates, then let advanceDownGreen rebuild it from the UdV chain
    m, s, n = inf.m, inf.s, inf.n
    for k in range(m, (n - 1) * s, -1):
        ctx.wrapDownGreen(k)
    Gw = ctx.g
    assert ctx.currentTimeslice == (n - 1) * s
    ctx.advanceDownGreen(n)
    Ga = ctx.g
    assert relerr(Gw, Ga) < 1e-7, "wrapped and re-stabilised G differ"
    # back to a consistent state
    ctx.setupUdVStorage_and_calculateGreen()
    assert relerr(ctx.g, G) < 1e-12
    # ONE updateInSlice at full size against the reference (round 3: ref_harness sliceTrace=2 -- updateInSliceThermalization(m)
    # and wrapDownGreen(m) after the 20 CPU SVDs of the construction; /root/reference/src/detsdwopdim.cpp:3023-3175, 3294-3375):
    # all 576 accept / reject decisions at n_g = 2304, the updated G and the wrapped G
    from dsfmt_oracle import RngWrapper
    op = oracle_params(g["params"]).finalize()
    r = RngWrapper(op.rngSeed, op.simindex + 1)
    for _ in range((op.opdim + 1) * op.N * op.m):           # the draws that went into the random field
        r.rand01()
    ctx.push_uniforms(np.array([r.rand01() for _ in range((op.opdim + 1) * op.N)]))
    ctx.updateInSlice(m, thermalization=True)
    st = ctx.update_state()
    phi_after = ctx.get_fields()[0]
    assert np.array_equal(phi_after[m], g["slice_phi_m"]), "accept / reject decisions at n_g = 2304 differ from the reference"
    assert abs(st.lastAccRatio - g["slice_accRatio"][0]) < 1e-15
    Gs = ctx.g
    assert relerr(Gs[::64, ::64], g["slice_g_sub64"]) < TOL and relerr(np.diag(Gs), g["slice_g_diag"]) < TOL
    assert abs(np.linalg.norm(Gs) - g["slice_g_fro"][0]) < TOL * g["slice_g_fro"][0]
    ctx.wrapDownGreen(m)
    Gw = ctx.g
    assert relerr(Gw[::64, ::64], g["slice_g_wrapped_sub64"]) < TOL and relerr(np.diag(Gw), g["slice_g_wrapped_diag"]) < TOL
    assert abs(np.linalg.norm(Gw) - g["slice_g_wrapped_fro"][0]) < TOL * g["slice_g_wrapped_fro"][0]
    # a fresh replica for the full thermalisation sweep (the slice above consumed uniforms outside the replica's RNG bookkeeping)
    rep.close()
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr"))
    ctx = rep.kernel_context
    rep.sweepThermalization()
    phi, ch, sh = ctx.get_fields()
    nrm = np.sqrt(np.sum(phi[1:] ** 2, axis=2))
    assert relerr(ch[1:], np.cosh(0.1 * nrm)) < 1e-14 and relerr(sh[1:], np.sinh(0.1 * nrm) / nrm) < 1e-13
    acc = rep.info.lastAccRatioLocal_phi
    assert 0.05 < acc < 0.99
    rng = np.random.default_rng(11)
    A = rng.standard_normal((2304, 64)) + 1j * rng.standard_normal((2304, 64))
    A = np.hstack([A] * 36)
    assert relerr(ctx.leftMultiplyBmatInv(ctx.leftMultiplyBmat(A, 20, 10), 20, 10), A) < 1e-10
    assert relerr(ctx.rightMultiplyBmat(ctx.rightMultiplyBmatInv(A, 200, 190), 200, 190), A) < 1e-10
    # after the down sweep storage[l] holds the L-type factors B(beta, l s): V_t = Q is unitary
    U, d, Vt = ctx.udv(3)
    assert np.all(d > 0)
    E = Vt.conj().T @ Vt - np.eye(2304)
    assert np.max(np.abs(E)) < 1e-11
    rep.close()


# ------------------------------------------------------------------------------------------------
# sub-batches: the chains of one handle spread over several kernel contexts swept by concurrent host threads
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["o2_L4_gshift", "o2_L4_wolffshift", "o2_L4_fmeas"])
def test_sub_batches_do_not_change_the_chains(name, tmp_path):
    """8 replicas in 4 kernel contexts x 2 chains (one host thread + HIP stream per context) against the same 8 replicas in
    ONE context: identical fields, step sizes, global-move statistics and observables -- and chain 0 on the reference's
    trajectory.  Also checkpoint / resume across the two layouts."""
    import dataclasses
    from detqmc_amd import DetSDWBatch
    g = load_golden(name)
    p0 = _sdw_params(g["params"], stabilisation="qr")
    plist = [dataclasses.replace(p0, simindex=p0.simindex + b, r=p0.r + 0.03 * b) for b in range(8)]
    one = DetSDWBatch(plist, sub_batches=1)
    four = DetSDWBatch(plist, sub_batches=4)
    assert one.sub_batches == 1 and four.sub_batches == 4 and len(four.kernel_contexts()) == 4
    nsw = 0
    while f"sweep{nsw + 1}_phi" in g:
        nsw += 1
        one.sweepThermalization()
        four.sweepThermalization()
        assert np.array_equal(four.chain(0).phi[1:], _golden_phi(g, f"sweep{nsw}_phi")[1:]), f"sweep {nsw}"
    assert nsw >= 1
    meas = "meas1_phi" in g
    if meas:
        one.sweep(True)
        four.sweep(True)
        assert np.array_equal(four.chain(0).phi[1:], _golden_phi(g, "meas1_phi")[1:])
    for b in range(8):
        a, c = one.chain(b), four.chain(b)
        assert np.array_equal(a.phi, c.phi), b
        assert relerr(c.g, a.g) < 1e-12, b
        ia, ic = a.info, c.info
        for f in ("phiDelta", "acceptedGlobalShifts", "attemptedGlobalShifts", "acceptedWolffClusterShiftUpdates", "addedWolffClusterSize",
                  "rngDrawn", "performedSweeps", "lastSweepDir", "currentTimeslice"):
            assert getattr(ia, f) == getattr(ic, f), (b, f)
        assert c.kernel_context.currentTimeslice == ic.currentTimeslice
        if meas:
            oa, oc = a.observables, c.observables
            assert oa.normMeanPhi == oc.normMeanPhi and oa.associatedEnergy == oc.associatedEnergy
            if oa.fermionic_valid:
                assert abs(oa.greenK0 - oc.greenK0) < 1e-10 * abs(oa.greenK0) and np.allclose(a.observable_vector("kOccX"), c.observable_vector("kOccX"), rtol=1e-10)
    # a checkpoint written by the 4-context layout resumes in the 1-context layout (even number of sweeps so far or not:
    # both continue from G(beta) rebuilt from the fields)
    ck = str(tmp_path / "state.bin")
    four.save_state(ck)
    one.load_state(ck)
    four.load_state(ck)
    one.sweepThermalization()
    four.sweepThermalization()
    for b in range(8):
        assert np.array_equal(one.chain(b).phi, four.chain(b).phi), b
    with pytest.raises(Exception):
        DetSDWBatch(plist, sub_batches=3)
    one.close()
    four.close()


def test_replica_exchange_across_two_processes_with_real_chains(tmp_path):
    """detqmc_amd/pt.py with REAL batched chains across two PROCESSES (torch.distributed, gloo; both ranks on this box's one
    GPU): 2 ranks x 2 chains must follow, step by step, the single-process run that holds all 4 chains in one batch --
    parameter indices, r on the device, step sizes, field configurations, and the observables filed per control parameter."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    rvalues, steps = [-1.2, -1.1, -1.0, -0.9], 4
    worker = os.path.join(ROOT, "tests", "pt_gpu_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, worker, str(tmp_path), json.dumps(rvalues), str(steps), "4"], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29500 + (os.getpid() % 400)), worker, str(tmp_path), json.dumps(rvalues), str(steps), "2"],
                         env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    ref = json.load(open(tmp_path / "rank0_of1.json"))
    got = [json.load(open(tmp_path / ("rank%d_of2.json" % r))) for r in range(2)]
    moved = False
    for p in range(4):
        for it in range(steps):
            a, b = ref["hist"][p][it], got[p // 2]["hist"][p % 2][it]
            assert a == b, (p, it, a, b)
            moved |= a["index"] != p
    assert moved, "ladder too steep for any swap"
    assert ref["proposed"] == got[0]["proposed"] and ref["accepted"] == got[0]["accepted"]
    assert np.allclose(np.array(ref["routed"]), np.array(got[0]["routed"]), rtol=1e-14, atol=0)


def test_decomposition_failure_is_reported_not_swallowed(monkeypatch):
    """udvDecompose throws "SVD failed" in the reference (src/udv.h:77-88); here the Jacobi SVD reports DQMC_ENOCONV through the
    C ABI when it cannot converge within its sweep budget (budget set to 1 through dqmc_tuning::max_jacobi_sweeps)."""
    from detqmc_amd import DqmcError, KernelContext, DetSDW, SDWParams
    ctx = KernelContext(2, 6, 20, 10, 0.1, delaySteps=4, stabilisation="svd", maxJacobiSweeps=1)
    rng = np.random.default_rng(1)
    M = rng.standard_normal((ctx.ng, ctx.ng)) + 1j * rng.standard_normal((ctx.ng, ctx.ng))
    with pytest.raises(DqmcError) as e:
        ctx.udvDecompose(M)
    assert e.value.code == -3 and "SVD failed" in str(e.value)
    ctx.close()
    with pytest.raises(DqmcError) as e2:           # the host layer turns it into the reference's GeneralError
        DetSDW(SDWParams(opdim=2, L=6, beta=2.0, s=10, stabilisation="svd", maxJacobiSweeps=1))
    assert e2.value.code == -3
    ctx = KernelContext(2, 6, 20, 10, 0.1, delaySteps=4, stabilisation="svd")
    U, d, Vt, sweeps = ctx.udvDecompose(M)
    assert sweeps > 1 and relerr((U * d[None, :]) @ Vt.conj().T, M) < 1e-12
    ctx.close()


@pytest.mark.parametrize("name", ["o2_L4_woodbury", "o3_L4_woodbury"])
@pytest.mark.parametrize("method", ["woodbury", "iterative"])
def test_immediate_update_methods_vs_reference_woodbury_run(name, method):
    """a19: the fixture is the reference run with updateMethod=woodbury (updateInSlice_woodbury, src/detsdwopdim.cpp:2885-3019;
    its updateMethod=iterative aborts with heap corruption in this build of the reference, see oracle/make_golden.py).  Here both
    immediate methods are the delayed kernels at delaySteps = 1; they must reproduce that run."""
    from detqmc_amd import DetSDW
    g = load_golden(name)
    assert g["params"]["updateMethod"] == "woodbury"
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", updateMethod=method))
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        i += 1
    assert relerr(rep.g, g[f"sweep{i - 1}_g"]) < TOL and rep.info.phiDelta == g[f"sweep{i - 1}_phiDelta"][0]
    assert np.array_equal([rep.rand01() for _ in range(4)], g["rng_next"])
    rep.close()


def test_device_phi_action_and_batched_transfers_vs_oracle():
    """phiAction (src/detsdwopdim.cpp:4242-4300) as a device reduction, the global displacement kernel and the one-transfer
    accessors of a batched context, against the oracle on the same fields (O(2) and O(3), every chain its own r)"""
    import dataclasses
    from detqmc_amd import DetSDWBatch
    from detsdw_oracle import DetSDWOracle
    for name in ("o2_L6_seed", "o3_L4"):
        g = load_golden(name)
        p0 = _sdw_params(g["params"], stabilisation="qr")
        plist = [dataclasses.replace(p0, simindex=p0.simindex + b, r=p0.r - 0.3 * b) for b in range(3)]
        batch = DetSDWBatch(plist, sub_batches=1)
        ctx = batch.kernel_context
        oras = []
        for b, p in enumerate(plist):
            op = oracle_params(g["params"])
            op.simindex, op.r = p.simindex, p.r
            oras.append(DetSDWOracle(op))
        fields = ctx.get_fields_all()
        for b, o in enumerate(oras):
            assert np.array_equal(fields[b][1:], o.phi[1:]), (name, b)
        act = ctx.phi_action_all()
        want = np.array([o.phiAction() for o in oras])
        assert np.all(np.abs(act - want) <= 1e-12 * np.abs(want)), (name, act, want)
        assert relerr(ctx.sv_all()[1], batch.chain(1).g_inv_sv) == 0
        shifts = np.array([[0.1 * (b + 1) * (d + 1) for d in range(p0.opdim)] for b in range(3)])
        ctx.shift_fields_all(shifts)
        for b, o in enumerate(oras):
            o.phi = o.phi + shifts[b][None, None, :]
            assert np.array_equal(batch.chain(b).phi, o.phi), (name, b)        # every slice incl. the unused slice 0
        act2 = ctx.phi_action_all()
        want2 = np.array([o.phiAction() for o in oras])
        assert np.all(np.abs(act2 - want2) <= 1e-12 * np.abs(want2))
        batch.close()


def test_lu_and_householder_green_functions_walk_the_same_chains(tmp_path):
    """greenFromUdV in QR mode inverts its scale-split matrix by LU with partial pivoting (kernels_lu.hip, n_g <= 512);
    greenVariant = 1 (dqmc_tuning::green_variant) keeps the Householder route of round 1 (and n_g > 512 uses it).  Both must give the
    chains the fixtures pin: same fields, same global-move decisions (log det from diag U resp. diag R), same G -- two child processes
    run scripts/check_lu_vs_qr.py (L = 8 with n_g = 128 = four 32-wide panels, L = 6 with n_g = 72: a ragged last one)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "check_lu_vs_qr.py")
    for L, beta, sweeps in ((8, 10.0, 6), (6, 20.0, 4)):
        outs = []
        for force_qr in (False, True):
            r = subprocess.run([sys.executable, script, str(L), str(beta), str(sweeps), "4"] + (["qr"] if force_qr else []),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            f = tmp_path / ("L%d_%s.txt" % (L, "qr" if force_qr else "lu"))
            f.write_text(r.stdout)
            outs.append(str(f))
        r = subprocess.run([sys.executable, script, "--compare"] + outs, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        assert "same chains" in r.stdout


@pytest.mark.parametrize("name", ["o2_L8_b5", "o3_L6", "o2_L6_seed"])
def test_pipelined_and_sequential_updates_walk_the_same_chain(name):
    """dqmc_update_slice can overlap the flush of a delayed-update block with the decisions of the next one (the decisions read a compact,
    already updated copy of their proposal window; dqmc_tuning::pipeline = 1, latched at dqmc_create); pipeline = -1 keeps the strictly
    sequential order.  Both must walk the chain of the reference fixture; between themselves: identical fields and RNG position, G to
    rounding.  dqmc_get_schedule_info tells which schedule really ran."""
    from detqmc_amd import DetSDW
    g = load_golden(name)
    out = []
    for pipeline in (1, -1):
        rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", pipeline=pipeline))
        i = 1
        while f"sweep{i}_phi" in g:
            rep.sweepThermalization()
            assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), (pipeline, i)
            if f"sweep{i}_g" in g:
                assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL
            else:                                   # larger fixtures keep a sub-sampled G and its diagonal
                assert relerr(np.diag(rep.g), g[f"sweep{i}_g_diag"]) < TOL
            i += 1
        si = rep.kernel_context.schedule_info()
        assert si.pipelined == (1 if pipeline == 1 else 0)
        assert (si.blocks_pipelined > 0 and si.blocks_sequential == 0) if pipeline == 1 else (si.blocks_pipelined == 0 and si.blocks_sequential > 0)
        out.append((rep.phi, rep.g, rep.info.rngDrawn))
        rep.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][2] == out[1][2]
    assert relerr(out[0][1], out[1][1]) < 1e-11


def test_config5_auto_pipelined_batch_slice_vs_reference():
    """The path `bench.py --config o3_L24_b20` times: a BATCH of config-5 replicas (n_g = 2304), for which dqmc_create switches the
    pipelined delayed update on by itself (dqmc_tuning::pipeline = 0: n_g > 1024 and at least two chains) -- flush of block b on the
    second stream, decisions of block b + 1 from the k_update_window copy (MSF = 4, 32-site window, 128 x 128 entries).  Chain 0 carries
    the fixture's parameters: all 576 accept / reject decisions of the reference's ONE updateInSlice at n_g = 2304, the updated G and
    the wrapped G (/root/reference/src/detsdwopdim.cpp:3023-3175, /root/reference/src/detmodel.h:1066-1095); dqmc_get_schedule_info
    proves that the pipelined branch was the one that ran (36 blocks, none sequential)."""
    import dataclasses
    from detqmc_amd import DetSDWBatch
    from dsfmt_oracle import RngWrapper
    g = load_golden("o3_L24_b20_init")
    p0 = _sdw_params(g["params"], stabilisation="qr")
    pars = [p0, dataclasses.replace(p0, simindex=p0.simindex + 1, r=p0.r - 0.05)]
    batch = DetSDWBatch(pars)
    ctx = batch.kernel_context
    si = ctx.schedule_info()
    assert si.pipelined == 1 and si.qr_block_gram_schmidt == 1 and si.blocks_pipelined == 0 and si.blocks_sequential == 0
    c0 = batch.chain(0)
    assert np.array_equal(_sha_phi(c0.phi), g["init_phi_sha256"])
    G = c0.g
    assert relerr(G[::64, ::64], g["init_g_sub64"]) < TOL and relerr(np.diag(G), g["init_g_diag"]) < TOL
    op = oracle_params(g["params"]).finalize()
    m = op.m
    for b, p in enumerate(pars):
        r = RngWrapper(p.rngSeed, p.simindex + 1)
        for _ in range((op.opdim + 1) * op.N * op.m):           # the draws that went into the random field
            r.rand01()
        ctx.select_chain(b)
        ctx.push_uniforms(np.array([r.rand01() for _ in range((op.opdim + 1) * op.N)]))
    ctx.updateInSlice(m, thermalization=True)
    si = ctx.schedule_info()
    assert si.blocks_pipelined == (op.N + op.delaySteps - 1) // op.delaySteps and si.blocks_sequential == 0
    ctx.select_chain(0)
    st = ctx.update_state()
    assert np.array_equal(ctx.get_fields()[0][m], g["slice_phi_m"]), "accept / reject decisions of the pipelined batch differ from the reference"
    assert abs(st.lastAccRatio - g["slice_accRatio"][0]) < 1e-15
    Gs = ctx.g
    assert relerr(Gs[::64, ::64], g["slice_g_sub64"]) < TOL and relerr(np.diag(Gs), g["slice_g_diag"]) < TOL
    assert abs(np.linalg.norm(Gs) - g["slice_g_fro"][0]) < TOL * g["slice_g_fro"][0]
    ctx.wrapDownGreen(m)
    Gw = ctx.g
    assert relerr(Gw[::64, ::64], g["slice_g_wrapped_sub64"]) < TOL and relerr(np.diag(Gw), g["slice_g_wrapped_diag"]) < TOL
    assert abs(np.linalg.norm(Gw) - g["slice_g_wrapped_fro"][0]) < TOL * g["slice_g_wrapped_fro"][0]
    # chain 1 went through the same launches with its own field and uniforms: its own single-replica twin, sequential schedule
    ctx.select_chain(1)
    phi1, G1 = ctx.get_fields()[0][m].copy(), ctx.g
    batch.close()
    from detqmc_amd import DetSDW
    twin = DetSDW(dataclasses.replace(pars[1], pipeline=-1))
    tctx = twin.kernel_context
    assert tctx.schedule_info().pipelined == 0
    r = RngWrapper(pars[1].rngSeed, pars[1].simindex + 1)
    for _ in range((op.opdim + 1) * op.N * op.m):
        r.rand01()
    tctx.push_uniforms(np.array([r.rand01() for _ in range((op.opdim + 1) * op.N)]))
    tctx.updateInSlice(m, thermalization=True)
    assert np.array_equal(tctx.get_fields()[0][m], phi1)
    tctx.wrapDownGreen(m)
    assert relerr(tctx.g, G1) < 1e-11
    twin.close()


def test_headline_size_pipelined_vs_reference_checksums():
    """The pipelined update schedule forced on at the headline size (dqmc_tuning::pipeline = 1, window copy of 64 sites x 2 bands)
    against the reference's fixture: fields after two sweeps bit-identical, G checksums 1e-10."""
    from detqmc_amd import DetSDW
    g = load_golden("o2_L16_b10")
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", delaySteps=32, pipeline=1))
    for i in (1, 2):
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}"
        G = rep.g
        assert relerr(G[::16, ::16], g[f"sweep{i}_g_sub16"]) < TOL and relerr(np.diag(G), g[f"sweep{i}_g_diag"]) < TOL
        assert rep.info.phiDelta == g[f"sweep{i}_phiDelta"][0]
    si = rep.kernel_context.schedule_info()
    assert si.pipelined == 1 and si.blocks_sequential == 0 and si.blocks_pipelined == 2 * 100 * 8
    rep.close()


@pytest.mark.parametrize("name", ["o2_L8_b5", "o3_L6", "o2_L8_apbc_flux", "o2_L4_gshift"])
def test_block_gram_schmidt_qr_walks_the_reference_chains(name):
    """The factorisation of config 5 (block Gram-Schmidt + Cholesky-QR2, automatic for n_g > 1024) forced on small lattices
    (dqmc_tuning::qr_variant = 2) together with the QR route of the Green's function (green_variant = 1, what n_g > 512 uses):
    same chains as the reference fixtures, no panel sent to the Householder fallback."""
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation="qr", qrVariant=2, greenVariant=1))
    si = rep.kernel_context.schedule_info()
    assert si.qr_block_gram_schmidt == 1 and si.green_lu == 0
    i = 1
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.phi[1:], _golden_phi(g, f"sweep{i}_phi")[1:]), f"sweep {i}: trajectory diverged"
        if f"sweep{i}_g" in g:
            assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL
        else:
            assert relerr(np.diag(rep.g), g[f"sweep{i}_g_diag"]) < TOL
        i += 1
    assert rep.kernel_context.schedule_info().cholqr_fallbacks == 0
    rep.close()


def test_cholesky_qr_failure_falls_back_to_householder_panels():
    """A panel too ill conditioned for Cholesky-QR (pivot test of k_chol64) is not an error and needs no restart: the factorisation
    is redone inside the same call with Householder panels (udt_dev, dqmc_context.hip) and counted.  Matrix: 128 x 128 with two
    column pairs that agree to 1e-9 (kappa of the column-equilibrated panel ~ 1e9, beyond what CholQR2 can orthogonalise).  The
    result must be as good as the Householder variant's own: U unitary to rounding, U d V^H = M to 1e-12."""
    from detqmc_amd import KernelContext
    rng = np.random.default_rng(5)
    n = 128
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    M[:, 17] = M[:, 3] + 1e-9 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    # equal norms: the norm pre-pivoting puts each pair side by side, so at least one pair shares a 64-column panel
    M[:, 90] = M[:, 77] * np.exp(0.7j) + 1e-9 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = {}
    for variant in (2, 1):
        ctx = KernelContext(2, 8, 20, 10, 0.1, delaySteps=8, stabilisation="qr", qrVariant=variant)
        U, d, Vt, _ = ctx.udvDecompose(M)
        si = ctx.schedule_info()
        assert si.qr_block_gram_schmidt == (1 if variant == 2 else 0)
        assert si.cholqr_fallbacks == (1 if variant == 2 else 0), "the ill-conditioned panel must be detected and redone"
        assert relerr(U.conj().T @ U, np.eye(n)) < 1e-12
        assert relerr((U * d[None, :]) @ Vt.conj().T, M) < 1e-12
        out[variant] = (U * d[None, :]) @ Vt.conj().T
        # a well-conditioned matrix right afterwards: no fallback, the flag was cleared
        M2 = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        U2, d2, Vt2, _ = ctx.udvDecompose(M2)
        assert ctx.schedule_info().cholqr_fallbacks == si.cholqr_fallbacks
        assert relerr((U2 * d2[None, :]) @ Vt2.conj().T, M2) < 1e-12 and relerr(U2.conj().T @ U2, np.eye(n)) < 1e-12
        ctx.close()
    assert relerr(out[2], out[1]) < 1e-12


@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", ["o3_L4_rotscale", "o3_L4_rotandscale", "o3_L6_rotscale_rep2", "o2_L4_rep3"])
def test_rotate_scale_proposals_and_repeated_updates_vs_reference(name, stab):
    """spinProposalMethod = rotate_then_scale / rotate_and_scale (O(3)) with the ADAPT_ROTATE / ADAPT_SCALE bisections, and
    repeatUpdateInSlice > 1, against the reference (/root/reference/src/detsdwopdim.cpp:2438-2470, 3934-4170, 3299-3375; the Box-Muller
    stack of src/normaldistribution.h).  Every accept / reject decision, the number of uniforms consumed (a Gaussian draw takes a
    variable number), angleDelta and scaleDelta after 12 sweeps are the reference's exactly.  The proposed vectors go through the
    device library's sincos / pow / log, which agree with glibc's to an ulp or two: rotated and scaled field values agree to 1e-13
    relative, not bit for bit (box proposals -- o2_L4_rep3 -- stay bit-identical)."""
    from detqmc_amd import DetSDW
    g = load_golden(name)
    rep = DetSDW(_sdw_params(g["params"], stabilisation=stab))
    assert np.array_equal(rep.phi[1:], _golden_phi(g, "init_phi")[1:])
    exact = g["params"].get("spinProposalMethod", "box") == "box"
    i = 1
    moved = set()
    while f"sweep{i}_phi" in g:
        rep.sweepThermalization()
        ref = _golden_phi(g, f"sweep{i}_phi")[1:]
        if exact:
            assert np.array_equal(rep.phi[1:], ref), f"sweep {i}: trajectory diverged"
        else:
            assert np.allclose(rep.phi[1:], ref, rtol=1e-13, atol=1e-15), f"sweep {i}: trajectory diverged"
        if f"sweep{i}_g" in g:
            assert relerr(rep.g, g[f"sweep{i}_g"]) < TOL, f"sweep {i}"
        else:
            assert relerr(np.diag(rep.g), g[f"sweep{i}_g_diag"]) < TOL, f"sweep {i}"
        inf = rep.info
        assert inf.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert abs(inf.lastAccRatioLocal_phi - g[f"sweep{i}_lastAccRatio"][0]) < 1e-15
        if f"sweep{i}_angleDelta" in g:
            assert inf.angleDelta == g[f"sweep{i}_angleDelta"][0] and inf.scaleDelta == g[f"sweep{i}_scaleDelta"][0], f"sweep {i}"
            moved.add((inf.angleDelta, inf.scaleDelta))
        i += 1
    if name == "o3_L4_rotscale":
        assert len(moved) >= 3
    nxt = np.array([rep.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"]), "RNG stream position differs from the reference"
    rep.close()


def test_rotate_scale_parameter_rules():
    """rotate / scale proposals exist for the O(3) model only (the reference throws from proposeRandomRotatedVector<OPDIM != 3>,
    /root/reference/src/detsdwopdim.cpp:3934-3942): rejected at create; dqmc_update_slice_ex rejects them for an O(2) context"""
    from detqmc_amd import DetSDW, DqmcError, SDWParams
    with pytest.raises(DqmcError) as e:
        DetSDW(SDWParams(opdim=2, L=4, beta=2.0, spinProposalMethod="rotate_then_scale"))
    assert "O(3)" in str(e.value)
    with pytest.raises(DqmcError):
        DetSDW(SDWParams(opdim=3, L=4, beta=2.0, repeatUpdateInSlice=-1))
    rep = DetSDW(SDWParams(opdim=2, L=4, beta=2.0))
    with pytest.raises(DqmcError) as e2:
        rep.kernel_context.updateInSlice(rep.info.m, proposal="rotate")
    assert "O(3)" in str(e2.value)
    rep.close()


def test_environment_cannot_change_the_markov_chain():
    """No environment variable may change what the decision kernel computes.  Rounds 1-2 shipped timing experiments behind
    DQMC_DBG (bit 2 skipped the p = W v / q = u W products and the bordering update, bit 4 replaced exp by 1 + x); they are gone,
    and the one developer switch left (phase timers, DQMC_DECIDE_TIMING) exists only in -DDQMC_DECIDE_TIMING builds.  A child
    process with those variables set must leave slice_phi_m / slice_g of the reference fixtures untouched (updateInSlice,
    /root/reference/src/detsdwopdim.cpp:3023-3175) -- and so must DQMC_SYNC_CHECK=1, the debug mode that waits for the stream
    at the end of every entry point."""
    import os
    import subprocess
    import sys
    if os.environ.get("DQMC_TEST_CHILD"):
        pytest.skip("already inside the child process")
    env = dict(os.environ, DQMC_DBG="6", DQMC_DECIDE_TIMING="1", DQMC_SYNC_CHECK="1", DQMC_TEST_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_update_slice_vs_reference and (o2_L8_b5 or o3_L4 or o2_L4_flux)"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "5 passed, 1 skipped" in r.stdout, r.stdout[-1000:]      # three fixtures x two launch shapes, O(3) has one shape
