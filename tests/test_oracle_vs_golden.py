"""Pins the oracle (oracle/detsdw_oracle.py, oracle/dsfmt_oracle.py) against fixtures generated
from the REAL reference build (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, oracle_params, relerr
from detsdw_oracle import DetSDWOracle, make_test_matrix
from dsfmt_oracle import RngWrapper

TOL = 1e-10   # BASELINE.json north_star: 1e-10 relative for fp64

SMALL = ["o2_L4", "o2_L4_s7", "o2_L4_flux", "o2_L4_apbc", "o1_L4", "o3_L4", "o2_L6_seed",
         "o2_L4_dense", "o2_L4_dense_flux",      # *_dense: checkerboard=false (CB_NONE)
         "o2_L4_cdw_slice", "o3_L4_cdw_slice"]   # cdwU != 0: the discrete field l_i(tau) and its second update pass (one slice)


def test_rng_bit_exact():
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "rng.npz"))
    for key in z.files:
        _, seed, pidx = key.split("_")
        r = RngWrapper(int(seed), int(pidx))
        mine = np.array([r.rand01() for _ in range(len(z[key]))])
        assert np.array_equal(mine, z[key]), key


# round 4: flux / antiperiodic boundaries at L = 8 (Landau-gauge phases for y = 0 .. 7, boundary-crossing plaquettes, CB_NONE + flux)
L8 = ["o2_L8_flux", "o2_L8_apbc_flux", "o2_L8_apbc", "o2_L8_dense_flux"]


@pytest.fixture(scope="module", params=SMALL + ["o2_L8_b5"] + L8)
def case(request):
    g = load_golden(request.param)
    o = DetSDWOracle(oracle_params(g["params"]))
    return request.param, g, o


def test_init_field_and_green(case):
    name, g, o = case
    # phi fixture is (N, OPDIM, m+1) col-major -> ours (m+1, N, OPDIM)
    phi_ref = np.transpose(g["init_phi"], (2, 0, 1))
    assert np.array_equal(o.phi[1:], phi_ref[1:]), "random field must be bit-identical (same RNG stream)"
    if "init_cdwl" in g:
        assert np.array_equal(o.cdwl[1:], g["init_cdwl"].T[1:])
    if "init_coshTermPhi" in g:
        assert relerr(o.coshTermPhi[1:], g["init_coshTermPhi"].T[1:]) < 1e-14
        assert relerr(o.sinhTermPhi[1:], g["init_sinhTermPhi"].T[1:]) < 1e-14
    assert relerr(o.g, g["init_g"]) < TOL
    assert relerr(o.g_inv_sv, g["init_g_inv_sv"]) < TOL
    d = np.stack([u.d for u in o.UdVStorage], axis=1)
    assert np.max(np.abs(d - g["init_udv_d"]) / g["init_udv_d"]) < 1e-9


def test_bmult(case):
    name, g, o = case
    if "bmult_k" not in g:
        pytest.skip("fixture keeps no B-multiply results")
    A = make_test_matrix(o.ng)
    k = int(g["bmult_k"][0])
    assert relerr(o.leftMultiplyBmat(A, k, k - 1), g["bmult_left"]) < 1e-13
    assert relerr(o.rightMultiplyBmat(A, k, k - 1), g["bmult_right"]) < 1e-13
    if "bmult_leftinv" in g:
        assert relerr(o.leftMultiplyBmatInv(A, k, k - 1), g["bmult_leftinv"]) < 1e-13
        assert relerr(o.rightMultiplyBmatInv(A, k, k - 1), g["bmult_rightinv"]) < 1e-13
    if "bchain_left" in g:
        k2 = int(g["bchain_k2"][0])
        assert relerr(o.leftMultiplyBmat(A, k2, 0), g["bchain_left"]) < 1e-12
        assert relerr(o.leftMultiplyBmatInv(A, k2, 0), g["bchain_leftinv"]) < 1e-12
        assert relerr(o.rightMultiplyBmat(A, k2, 0), g["bchain_right"]) < 1e-12
        assert relerr(o.rightMultiplyBmatInv(A, k2, 0), g["bchain_rightinv"]) < 1e-12
    if "bdense_k" in g:
        assert relerr(o.computeBmatDense(k), g["bdense_k"]) < 1e-12


def test_slice_and_sweeps(case):
    """Follows the trace the harness took: one slice of delayed updates at k=m, wrap, the rest of the
    down sweep, then full sweeps.  Same RNG stream => same Markov chain."""
    name, g, o = case
    m, n, s = o.m, o.n, o.s
    o.updateInSliceThermalization(m)
    assert np.array_equal(o.phi[m], g["slice_phi_m"]), "accept/reject decisions must agree"
    if "slice_cdwl_m" in g:
        assert np.array_equal(o.cdwl[m], g["slice_cdwl_m"].reshape(-1)), "accept/reject decisions of the cdwl pass must agree"
    assert relerr(o.g, g["slice_g"]) < TOL
    assert abs(o.lastAccRatioLocal_phi - g["slice_accRatio"][0]) < 1e-15
    o.wrapDownGreen(m)
    if "slice_g_wrapped" in g:
        assert relerr(o.g, g["slice_g_wrapped"]) < TOL
    if "adv_g" not in g:          # sliceTrace=2 fixtures end here
        return
    for k in range(m - 1, (n - 1) * s, -1):
        o.updateInSliceThermalization(k)
        o.wrapDownGreen(k)
    for l in range(n - 1, 0, -1):
        o.advanceDownGreen(l + 1)
        if l == n - 1:
            assert relerr(o.g, g["adv_g"]) < TOL
            assert relerr(o.g_inv_sv, g["adv_g_inv_sv"]) < TOL
        for k in range(l * s, (l - 1) * s, -1):
            o.updateInSliceThermalization(k)
            o.wrapDownGreen(k)
    o.advanceDownGreen(1)
    o.lastSweepDir = -1
    o.performedSweeps += 1
    i = 1
    while f"sweep{i}_phi" in g:
        if i > 1:
            o.sweepThermalization()
        phi_ref = np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))
        assert np.array_equal(o.phi[1:], phi_ref[1:]), f"sweep {i}: field trajectory diverged"
        if f"sweep{i}_cdwl" in g:
            assert np.array_equal(o.cdwl[1:], g[f"sweep{i}_cdwl"].T[1:]), f"sweep {i}: cdwl trajectory diverged"
        assert relerr(o.g, g[f"sweep{i}_g"]) < TOL
        assert relerr(o.g_inv_sv, g[f"sweep{i}_g_inv_sv"]) < TOL
        assert o.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert abs(o.lastAccRatioLocal_phi - g[f"sweep{i}_lastAccRatio"][0]) < 1e-15
        i += 1
    assert abs(o.get_exchange_action_contribution() - g["exchange_action"][0]) < 1e-12 * abs(g["exchange_action"][0])
    nxt = np.array([o.rng.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"]), "number of RNG draws consumed differs from the reference"


@pytest.mark.parametrize("name", ["o2_L4_cdw", "o1_L4_cdw", "o2_L4_cdw_gshift", "o2_L4_cdw_dense"])
def test_cdw_trajectory(name):
    """cdwU != 0 over whole sweeps (with global shift moves, a flux, the dense propagator): phi, the discrete field, G and the number
    of uniforms consumed.  Seeds: oracle/find_cdw_seeds.py (the reference's last-bit branch at null cdwl proposals)."""
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    assert np.array_equal(o.cdwl[1:], g["init_cdwl"].T[1:])
    assert relerr(o.g, g["init_g"]) < TOL
    i = 1
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        assert np.array_equal(o.phi[1:], np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))[1:]), f"sweep {i}"
        assert np.array_equal(o.cdwl[1:], g[f"sweep{i}_cdwl"].T[1:]), f"sweep {i}"
        assert relerr(o.g, g[f"sweep{i}_g"]) < TOL
        assert o.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert o.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    assert i > 1
    assert np.array_equal(np.array([o.rng.rand01() for _ in range(4)]), g["rng_next"])


def test_global_shift_trajectory():
    g = load_golden("o2_L4_gshift")
    o = DetSDWOracle(oracle_params(g["params"]))
    i = 1
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        phi_ref = np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))
        assert np.array_equal(o.phi[1:], phi_ref[1:]), f"sweep {i}"
        assert relerr(o.g, g[f"sweep{i}_g"]) < TOL
        assert o.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert o.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        i += 1
    assert o.attemptedGlobalShifts >= 2
    nxt = np.array([o.rng.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])


@pytest.mark.parametrize("name", ["o2_L4_wolff", "o2_L4_wolffshift", "o1_L4_wolff", "o3_L4_wolff"])
def test_wolff_cluster_moves_trajectory(name):
    """attemptWolffClusterUpdate / attemptWolffClusterShiftUpdate / buildAndFlipCluster (detsdwopdim.cpp:3488-3883)"""
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    i = 1
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        phi_ref = np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))
        assert np.array_equal(o.phi[1:], phi_ref[1:]), f"sweep {i}"
        assert relerr(o.g, g[f"sweep{i}_g"]) < TOL
        assert o.attemptedGlobalShifts == int(g[f"sweep{i}_attGlobalShifts"][0])
        assert o.acceptedGlobalShifts == int(g[f"sweep{i}_accGlobalShifts"][0])
        assert o.attemptedWolffClusterUpdates == int(g[f"sweep{i}_attWolff"][0])
        assert o.acceptedWolffClusterUpdates == int(g[f"sweep{i}_accWolff"][0])
        assert o.attemptedWolffClusterShiftUpdates == int(g[f"sweep{i}_attWolffShift"][0])
        assert o.acceptedWolffClusterShiftUpdates == int(g[f"sweep{i}_accWolffShift"][0])
        assert o.addedWolffClusterSize == g[f"sweep{i}_addedWolffClusterSize"][0]
        i += 1
    assert o.attemptedWolffClusterUpdates + o.attemptedWolffClusterShiftUpdates >= 2
    assert o.acceptedWolffClusterUpdates + o.acceptedWolffClusterShiftUpdates >= 1, "fixture exercises no accepted cluster move"
    nxt = np.array([o.rng.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"])


@pytest.mark.parametrize("name", ["o2_L4_meas", "o3_L4_meas"])
def test_measurement_sweeps(name):
    """sweep(true): bosonic observables of measure / finishMeasurements (detsdwopdim.cpp:509-545, :903-921), bit for bit"""
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    i = 1
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        i += 1
    j = 1
    while f"meas{j}_phi" in g:
        o.sweep(True)
        assert np.array_equal(o.phi[1:], np.transpose(g[f"meas{j}_phi"], (2, 0, 1))[1:])
        assert o.phiDelta == g[f"meas{j}_phiDelta"][0]
        assert np.array_equal(o.meanPhi, g[f"meas{j}_meanPhi"].ravel())
        assert o.normMeanPhi == g[f"meas{j}_normMeanPhi"][0]
        assert o.associatedEnergy == g[f"meas{j}_associatedEnergy"][0]
        if o.OPDIM == 2:
            assert o.phiRhoS_Gc == g[f"meas{j}_phiRhoS_Gc"][0] and o.phiRhoS_Gs == g[f"meas{j}_phiRhoS_Gs"][0]
        j += 1
    assert j > 2


@pytest.mark.parametrize("name", ["o2_L4_fmeas", "o2_L4_fmeas_apbc_flux", "o3_L4_fmeas", "o1_L4_fmeas", "o2_L8_fmeas_apbc_flux"])
def test_fermionic_measurements(name):
    """shiftGreenSymmetric (detsdwopdim.cpp:4507-4612) and the G-dependent observables of measure / finishMeasurements
    (:545-899, :923-1015): greenK0, greenLocal, k-space occupation, pairing correlators, occDiffSq"""
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    i = 1
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        i += 1
    j = 1
    while f"meas{j}_phi" in g:
        o.sweep(True)
        assert np.array_equal(o.phi[1:], np.transpose(g[f"meas{j}_phi"], (2, 0, 1))[1:])
        for key in ("greenK0", "greenLocal", "pairPlusMax", "pairMinusMax", "occDiffSq"):
            assert abs(getattr(o, key) - g[f"meas{j}_{key}"][0]) < TOL * max(1.0, abs(g[f"meas{j}_{key}"][0])), key
        for key in ("kOccX", "kOccY", "pairPlus", "pairMinus"):
            assert relerr(getattr(o, key), g[f"meas{j}_{key}"].ravel()) < TOL, key
        j += 1
    assert relerr(o.shiftGreenSymmetric(), g["final_shiftGreenSymmetric"]) < 1e-12


def test_headline_size_checksums():
    """BASELINE config 3 (L=16, beta=10): the reference's G is pinned through sub-samples, its
    diagonal, Frobenius norm and singular values; the field trajectory must be identical."""
    import time
    g = load_golden("o2_L16_b10")
    t0 = time.time()
    o = DetSDWOracle(oracle_params(g["params"]))
    assert relerr(o.g[::16, ::16], g["init_g_sub16"]) < TOL
    assert relerr(np.diag(o.g), g["init_g_diag"]) < TOL
    assert abs(np.linalg.norm(o.g) - g["init_g_fro"][0]) < TOL * g["init_g_fro"][0]
    assert relerr(o.g_inv_sv, g["init_g_inv_sv"]) < TOL
    o.sweepThermalization()
    phi_ref = np.transpose(g["sweep1_phi"], (2, 0, 1))
    assert np.array_equal(o.phi[1:], phi_ref[1:])
    assert relerr(o.g[::16, ::16], g["sweep1_g_sub16"]) < TOL
    assert relerr(np.diag(o.g), g["sweep1_g_diag"]) < TOL
    assert relerr(o.g_inv_sv, g["sweep1_g_inv_sv"]) < TOL
    print("oracle L=16: init + 1 sweep in %.1f s" % (time.time() - t0))


@pytest.mark.parametrize("name", ["o2_L8_b20", "o3_L6"])
def test_cold_and_large_o3_cases(name):
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    assert relerr(o.g[::4, ::4], g["init_g_sub4"]) < TOL
    assert relerr(np.diag(o.g), g["init_g_diag"]) < TOL
    o.sweepThermalization()
    phi_ref = np.transpose(g["sweep1_phi"], (2, 0, 1))
    assert np.array_equal(o.phi[1:], phi_ref[1:])
    assert relerr(o.g[::4, ::4], g["sweep1_g_sub4"]) < TOL
    assert relerr(o.g_inv_sv, g["sweep1_g_inv_sv"]) < TOL


# ------------------------------------------------------------------------------------------------
# Hubbard replica (BASELINE config 1): oracle/dethubbard_oracle.py against the real reference
# ------------------------------------------------------------------------------------------------
def _hubbard_params(a):
    from dethubbard_oracle import HubbardParams
    return HubbardParams(L=int(a["L"]), d=int(a["d"]), beta=float(a["beta"]), dtau=float(a["dtau"]), s=int(a["s"]), t=float(a["t"]),
                         U=float(a["U"]), mu=float(a["mu"]), checkerboard=bool(int(a["checkerboard"])),
                         rngSeed=int(a.get("rngSeed", 1020304050)), simindex=int(a.get("simindex", 0)))


@pytest.mark.parametrize("name", ["hub_L4", "hub_L4_cb", "hub_L4_s7", "hub_L6"])
def test_hubbard_oracle_vs_reference(name):
    from dethubbard_oracle import DetHubbardOracle
    g = load_golden(name)
    o = DetHubbardOracle(_hubbard_params(g["params"]))
    assert (o.N, o.m, o.s, o.n) == tuple(int(x) for x in g["meta"][1:5]) and abs(o.alpha - g["meta"][5]) < 1e-15
    assert relerr(o.proptmat, g["proptmat"]) < 1e-13
    assert np.array_equal(o.auxfield[:, 1:], g["init_auxfield"][:, 1:])
    assert relerr(o.g[0], g["init_gUp"]) < 1e-10 and relerr(o.g[1], g["init_gDn"]) < 1e-10
    assert relerr(o.computeBmat(3, 2, o.UP), g["bmat_up_k3"]) < 1e-13
    assert relerr(o.computeBmat(min(o.s, o.m), 0, o.DN), g["bmat_dn_chain"]) < 1e-12
    assert relerr(np.column_stack([st.d for st in o.storage[0]]), g["init_udv_d_up"]) < 1e-10
    i = 1
    while f"sweep{i}_auxfield" in g:
        o.sweepThermalization()
        assert np.array_equal(o.auxfield[:, 1:], g[f"sweep{i}_auxfield"][:, 1:]), f"sweep {i}: auxiliary field trajectory diverged"
        assert relerr(o.g[0], g[f"sweep{i}_gUp"]) < 1e-10 and relerr(o.g[1], g[f"sweep{i}_gDn"]) < 1e-10
        i += 1
    i = 1
    while f"meas{i}_auxfield" in g:
        o.sweep(True)
        assert np.array_equal(o.auxfield[:, 1:], g[f"meas{i}_auxfield"][:, 1:])
        got = [o.obs[k] for k in ("occUp", "occDn", "occTotal", "occDouble", "localMoment", "eKinetic", "ePotential", "eTotal")]
        assert np.allclose(got, g[f"meas{i}_obs"], rtol=1e-10, atol=1e-12)
        assert np.max(np.abs(o.zcorr - g[f"meas{i}_zcorr"])) < 1e-8        # sums of products of G entries: absolute, at the scale of G
        i += 1
    assert np.array_equal([o.rng.rand01() for _ in range(4)], g["rng_next"])
    if name == "hub_L4":        # known answer: half filling at mu = 0
        assert abs(o.obs["occTotal"] - 1.0) < 1e-9


@pytest.mark.parametrize("name", ["o3_L4_rotscale", "o3_L4_rotandscale", "o3_L6_rotscale_rep2", "o2_L4_rep3"])
def test_rotate_scale_proposals_and_repeated_updates(name):
    """spinProposalMethod = rotate_then_scale / rotate_and_scale with ADAPT_ROTATE / ADAPT_SCALE (O(3); detsdwopdim.cpp:2438-2470,
    3934-4170, 3299-3375; Box-Muller stack of normaldistribution.h) and repeatUpdateInSlice > 1: same libm as the reference in this
    container, so the field is bit-identical and angleDelta / scaleDelta move exactly as the reference's do."""
    g = load_golden(name)
    o = DetSDWOracle(oracle_params(g["params"]))
    assert np.array_equal(o.phi[1:], np.transpose(g["init_phi"], (2, 0, 1))[1:])
    i = 1
    moved = set()
    while f"sweep{i}_phi" in g:
        o.sweepThermalization()
        assert np.array_equal(o.phi[1:], np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))[1:]), f"sweep {i}: field trajectory diverged"
        if f"sweep{i}_g" in g:
            assert relerr(o.g, g[f"sweep{i}_g"]) < TOL
        else:
            assert relerr(np.diag(o.g), g[f"sweep{i}_g_diag"]) < TOL
        assert o.phiDelta == g[f"sweep{i}_phiDelta"][0]
        assert abs(o.lastAccRatioLocal_phi - g[f"sweep{i}_lastAccRatio"][0]) < 1e-15
        if f"sweep{i}_angleDelta" in g:
            assert o.angleDelta == g[f"sweep{i}_angleDelta"][0] and o.scaleDelta == g[f"sweep{i}_scaleDelta"][0]
            moved.add((o.angleDelta, o.scaleDelta))
        i += 1
    if name == "o3_L4_rotscale":
        assert len(moved) >= 3, "the fixture must exercise both adaptations"
    nxt = np.array([o.rng.rand01() for _ in range(4)])
    assert np.array_equal(nxt, g["rng_next"]), "number of RNG draws consumed differs from the reference"
