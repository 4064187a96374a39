"""GPU parity of the Hubbard replica (BASELINE config 1) through the C ABI: the HIP path against fixtures
produced by the REAL reference (src/dethubbard.cpp via oracle/ref_build/ref_harness_hubbard.cpp) and against the numpy
oracle.  Auxiliary-field trajectories must agree exactly (same Markov chain), Green's functions to 1e-10."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
CASES = ["hub_L4", "hub_L4_cb", "hub_L4_s7", "hub_L6"]


def _params(a, **over):
    from detqmc_amd import HubbardParams
    kw = dict(L=int(a["L"]), d=int(a["d"]), beta=float(a["beta"]), dtau=float(a["dtau"]), s=int(a["s"]), t=float(a["t"]), U=float(a["U"]),
              mu=float(a["mu"]), checkerboard=bool(int(a["checkerboard"])), rngSeed=int(a.get("rngSeed", 1020304050)),
              simindex=int(a.get("simindex", 0)))
    kw.update(over)
    return HubbardParams(**kw)


@pytest.mark.parametrize("stab", ["svd", "qr"])
@pytest.mark.parametrize("name", CASES)
def test_hubbard_replica_trajectory_vs_reference(name, stab):
    from detqmc_amd import DetHubbard
    g = load_golden(name)
    rep = DetHubbard(_params(g["params"], stabilisation=stab))
    inf = rep.info
    assert (inf.N, inf.m, inf.s, inf.n) == tuple(int(x) for x in g["meta"][1:5]) and abs(inf.alpha - g["meta"][5]) < 1e-15
    assert np.array_equal(rep.auxfield[:, 1:], g["init_auxfield"][:, 1:])
    gu, gd = rep.green
    assert relerr(gu, g["init_gUp"]) < 1e-10 and relerr(gd, g["init_gDn"]) < 1e-10
    i = 1
    while f"sweep{i}_auxfield" in g:
        rep.sweepThermalization()
        assert np.array_equal(rep.auxfield[:, 1:], g[f"sweep{i}_auxfield"][:, 1:]), f"sweep {i}: auxiliary field trajectory diverged"
        gu, gd = rep.green
        assert relerr(gu, g[f"sweep{i}_gUp"]) < 1e-10 and relerr(gd, g[f"sweep{i}_gDn"]) < 1e-10, f"sweep {i}"
        i += 1
    i = 1
    while f"meas{i}_auxfield" in g:
        rep.sweep(True)
        assert np.array_equal(rep.auxfield[:, 1:], g[f"meas{i}_auxfield"][:, 1:])
        o = rep.observables
        got = [o.occUp, o.occDn, o.occTotal, o.occDouble, o.localMoment, o.eKinetic, o.ePotential, o.eTotal]
        assert np.allclose(got, g[f"meas{i}_obs"], rtol=1e-10, atol=1e-12), f"measurement sweep {i}"
        assert np.max(np.abs(rep.zcorr - g[f"meas{i}_zcorr"])) < 1e-8
        i += 1
    assert np.array_equal([rep.rand01() for _ in range(4)], g["rng_next"])
    if name == "hub_L4":                     # known answer: half filling at mu = 0
        assert abs(rep.observables.occTotal - 1.0) < 1e-9
    rep.close()


def test_hubbard_b_matrix_and_block_structure_vs_reference():
    """B(k2, k1) of both spin sectors through the kernel ABI (dense propagator GEMM + site-diagonal factor) and its exact
    inverse; the two sectors never mix"""
    from detqmc_amd import DetHubbard, KernelContext
    from detqmc_amd.model import _CtxView
    g = load_golden("hub_L4")
    rep = DetHubbard(_params(g["params"]))
    inf = rep.info

    class _I:        # what _CtxView reads
        opdim, L, m, s, N, MSF, n_g, n = 1, inf.L, inf.m, inf.s, inf.N, 2, 2 * inf.N, inf.n
    ctx = _CtxView(rep.lib, rep.lib.dethubbard_ctx(rep.h), _I)
    N, ng = inf.N, 2 * inf.N
    eye = np.eye(ng)
    B = ctx.leftMultiplyBmat(eye, 3, 2)
    assert relerr(B[:N, :N].real, g["bmat_up_k3"]) < 1e-13
    assert np.max(np.abs(B[:N, N:])) == 0 and np.max(np.abs(B[N:, :N])) == 0 and np.max(np.abs(B.imag)) == 0
    Bc = ctx.leftMultiplyBmat(eye, min(inf.s, inf.m), 0)
    assert relerr(Bc[N:, N:].real, g["bmat_dn_chain"]) < 1e-12
    assert relerr(ctx.rightMultiplyBmat(eye, min(inf.s, inf.m), 0), Bc) < 1e-13
    rng = np.random.default_rng(3)
    A = rng.standard_normal((ng, ng))
    assert relerr(ctx.leftMultiplyBmatInv(ctx.leftMultiplyBmat(A, 7, 2), 7, 2), A) < 1e-11
    assert relerr(ctx.rightMultiplyBmat(ctx.rightMultiplyBmatInv(A, 20, 15), 20, 15), A) < 1e-11
    Gfull = ctx.g
    assert np.max(np.abs(Gfull[:N, N:])) < 1e-14 and np.max(np.abs(Gfull.imag)) == 0
    rep.close()


def test_hubbard_batched_replicas_and_oracle():
    """3 replicas in lockstep: replica 0 is the fixture's chain, the others follow the numpy oracle with their own streams"""
    import dataclasses
    from detqmc_amd import DetHubbard
    from dethubbard_oracle import DetHubbardOracle, HubbardParams as OP
    g = load_golden("hub_L4_s7")
    p0 = _params(g["params"], stabilisation="qr")
    rep = DetHubbard(p0, nchains=3)
    a = g["params"]
    oras = [DetHubbardOracle(OP(L=p0.L, d=2, beta=p0.beta, dtau=p0.dtau, s=p0.s, t=p0.t, U=p0.U, mu=p0.mu, checkerboard=p0.checkerboard,
                                rngSeed=p0.rngSeed, simindex=p0.simindex + b)) for b in range(3)]
    for i in (1, 2):
        rep.sweepThermalization()
        for o in oras:
            o.sweepThermalization()
        rep.select(0)
        assert np.array_equal(rep.auxfield[:, 1:], g[f"sweep{i}_auxfield"][:, 1:])
        for b, o in enumerate(oras):
            rep.select(b)
            assert np.array_equal(rep.auxfield[:, 1:], o.auxfield[:, 1:]), (i, b)
            gu, gd = rep.green
            assert relerr(gu, o.g[0]) < 1e-10 and relerr(gd, o.g[1]) < 1e-10
    rep.close()


def test_hubbard_parameter_rules():
    from detqmc_amd import DetHubbard, DqmcError, HubbardParams
    with pytest.raises(DqmcError):
        DetHubbard(HubbardParams(L=4, beta=2.0, m=20))            # only one of beta and m
    with pytest.raises(DqmcError):
        DetHubbard(HubbardParams(L=4, d=3, beta=2.0))             # this build: d = 2
    with pytest.raises(DqmcError):
        DetHubbard(HubbardParams(L=3, beta=2.0, checkerboard=True))
