"""Thin Python mirrors of the two native layers.

  KernelContext  <-> dqmc_ctx (include/dqmc_hip.h): the kernel-level ABI, method names follow the
                     reference functions each call replaces (detmodel.h / detsdwopdim.cpp).
  DetSDW         <-> detqmc::DetSDW (C++ host layer, include/detsdw_host.h): the replica with the
                     reference's operator surface: sweep(), sweepThermalization(),
                     get/set_exchange_parameter_value(), get_exchange_action_contribution(), ...

numpy arrays cross the boundary in the reference's layouts: complex128 column-major n_g x n_g
matrices, phi as (N, OPDIM, m+1) column-major.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, load

BC = {"pbc": 0, "apbc-x": 1, "apbc-y": 2, "apbc-xy": 3}
UPDATE_METHOD = {"iterative": 0, "woodbury": 1, "delayed": 2}
STABILISATION = {"svd": 0, "qr": 1}
LEFT, RIGHT = 0, 1
UP, DOWN = +1, -1


@dataclass
class SDWParams:
    """ModelParamsDetSDW (reference src/detsdwparams.h:24-120), defaults of the shipped examples."""
    opdim: int = 2
    L: int = 4
    beta: float = 0.0
    m: int = 0
    dtau: float = 0.1
    s: int = 10
    r: float = -1.0
    c: float = 3.0
    u: float = 1.0
    lambda_: float = 1.0
    txhor: float = -1.0
    txver: float = -0.5
    tyhor: float = 0.5
    tyver: float = 1.0
    mu: float = -0.5
    mux: float = None
    muy: float = None
    accRatio: float = 0.5
    delaySteps: int = 16
    updateMethod: str = "delayed"
    bc: str = "pbc"
    weakZflux: bool = False
    globalShift: bool = False
    wolffClusterUpdate: bool = False
    wolffClusterShiftUpdate: bool = False
    repeatWolffPerSweep: int = 1
    fermionMeasurements: bool = False    # sweep(True) also takes the G-dependent observables (reference default: on)
    globalUpdateInterval: int = 100
    phi2bosons: bool = False
    cdwU: float = 0.0
    rngSeed: int = 1020304050
    simindex: int = 0
    device: int = 0
    stabilisation: str = "svd"   # "svd": UdV = SVD like the reference; "qr": pre-pivoted Householder UDT
    checkerboard: bool = True    # False = CB_NONE: dense B_k = e^{-dtau V_k} e^{-dtau K} (reference option checkerboard=false)
    spinProposalMethod: str = "box"      # "box", "rotate_then_scale", "rotate_and_scale" (the latter two: opdim = 3 only)
    adaptScaleVariance: bool = False
    repeatUpdateInSlice: int = 1
    # result-neutral execution choices (dqmc_tuning, include/dqmc_hip.h); 0 = automatic
    pipeline: int = 0            # 1 / -1: pipelined delayed updates on / off
    qrVariant: int = 0           # 1: Householder panels, 2: block Gram-Schmidt + Cholesky-QR2
    greenVariant: int = 0        # 1: QR instead of LU inside greenFromUdV
    maxJacobiSweeps: int = 0     # SVD mode: sweep budget of the Jacobi SVD (0 = 80)
    proposalBudget: int = 0      # proposals per delayed-update block (-1: no limit)
    decideThreads: int = 0       # threads per workgroup of the decision kernel (0: automatic, 256, 512); launch shape only


SPIN_PROPOSAL = {"box": 0, "rotate_then_scale": 1, "rotate_and_scale": 2}
PROPOSE = {"box": 0, "rotate": 1, "scale": 2, "rotate_and_scale": 3}
ADAPT = {"box": 0, "rotate": 1, "scale": 2}


def _tuning(pipeline=0, qrVariant=0, greenVariant=0, maxJacobiSweeps=0, proposalBudget=0, decideThreads=0):
    return _lib.dqmc_tuning(pipeline=int(pipeline), qr_variant=int(qrVariant), green_variant=int(greenVariant),
                            max_jacobi_sweeps=int(maxJacobiSweeps), proposal_budget=int(proposalBudget),
                            decide_threads=int(decideThreads))


def _fmat(a):
    """complex128, column-major, owned."""
    return np.array(a, dtype=np.complex128, order="F", copy=True)


class KernelContext:
    """One dqmc_ctx.  Needs a GPU."""

    def __init__(self, opdim, L, m, s, dtau, delaySteps=16, bc="pbc", weakZflux=False, r=-1.0, c=3.0, u=1.0,
                 lambda_=1.0, txhor=-1.0, txver=-0.5, tyhor=0.5, tyver=1.0, mux=-0.5, muy=-0.5,
                 accRatio=0.5, phi2bosons=False, device=0, stabilisation="svd", checkerboard=True, nchains=1, cdwU=0.0,
                 pipeline=0, qrVariant=0, greenVariant=0, maxJacobiSweeps=0, proposalBudget=0, rngWindowPerSite=0, decideThreads=0):
        self.lib = load()
        p = _lib.dqmc_params(opdim=opdim, L=L, m=m, s=s, delaySteps=delaySteps, bc=BC[bc],
                             weakZflux=int(weakZflux), phi2bosons=int(phi2bosons), device=device,
                             stabilisation=STABILISATION[stabilisation], cb_none=int(not checkerboard), dtau=dtau, r=r, c=c, u=u, lambda_=lambda_, txhor=txhor, txver=txver,
                             tyhor=tyhor, tyver=tyver, mux=mux, muy=muy, accRatio=accRatio, cdwU=cdwU, rng_window_per_site=int(rngWindowPerSite),
                             tuning=_tuning(pipeline, qrVariant, greenVariant, maxJacobiSweeps, proposalBudget, decideThreads))
        h = C.c_void_p()
        check(self.lib.dqmc_create_batch(C.byref(p), nchains, C.byref(h)))
        self.h = h
        self.nchains = nchains
        self.opdim, self.L, self.m, self.s = opdim, L, m, s
        self.N = L * L
        self.MSF = 4 if opdim == 3 else 2
        self.ng = self.MSF * self.N
        self.n = -(-m // s)

    def close(self):
        if self.h:
            self.lib.dqmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shiftGreenSymmetric(self):
        g = np.zeros((self.ng, self.ng), dtype=np.complex128, order="F")
        check(self.lib.dqmc_shift_green_symmetric_host(self.h, g.ctypes.data))
        return g

    def select_chain(self, b):
        """host-buffer calls (fields, G, sv, UdV, uniforms, update state, ...) refer to chain b from now on"""
        check(self.lib.dqmc_select_chain(self.h, b))

    # fields: phi given/returned as (m+1, N, OPDIM) [oracle layout]; the ABI uses (N, OPDIM, m+1) col-major
    def set_fields(self, phi_kNd):
        ref = np.asfortranarray(np.transpose(np.asarray(phi_kNd, dtype=np.float64), (1, 2, 0)))
        check(self.lib.dqmc_set_fields_host(self.h, ref.ctypes.data_as(_lib._DP)))

    # the discrete field of cdwU != 0: (m+1, N) int32, values +-1 / +-2 (slice 0 unused)
    def set_cdwl(self, cdwl_kN):
        a = np.ascontiguousarray(cdwl_kN, dtype=np.int32)
        assert a.shape == (self.m + 1, self.N)
        check(self.lib.dqmc_set_cdwl_host(self.h, a.ctypes.data))

    def get_cdwl(self):
        a = np.zeros((self.m + 1, self.N), dtype=np.int32)
        check(self.lib.dqmc_get_cdwl_host(self.h, a.ctypes.data))
        return a

    def get_fields(self):
        phi = np.zeros((self.N, self.opdim, self.m + 1), order="F")
        ch = np.zeros((self.N, self.m + 1), order="F")
        sh = np.zeros((self.N, self.m + 1), order="F")
        check(self.lib.dqmc_get_fields_host(self.h, phi.ctypes.data_as(_lib._DP), ch.ctypes.data_as(_lib._DP),
                                            sh.ctypes.data_as(_lib._DP)))
        return np.transpose(phi, (2, 0, 1)).copy(), ch.T.copy(), sh.T.copy()

    def bmult(self, side, inverse, k2, k1, A):
        a = _fmat(A)
        check(self.lib.dqmc_bmult_host(self.h, side, int(inverse), k2, k1, a.ctypes.data))
        return a

    def leftMultiplyBmat(self, A, k2, k1):
        return self.bmult(LEFT, 0, k2, k1, A)

    def leftMultiplyBmatInv(self, A, k2, k1):
        return self.bmult(LEFT, 1, k2, k1, A)

    def rightMultiplyBmat(self, A, k2, k1):
        return self.bmult(RIGHT, 0, k2, k1, A)

    def rightMultiplyBmatInv(self, A, k2, k1):
        return self.bmult(RIGHT, 1, k2, k1, A)

    def udvDecompose(self, M):
        a = _fmat(M)
        U = np.zeros_like(a)
        Vt = np.zeros_like(a)
        d = np.zeros(self.ng)
        sw = C.c_int(0)
        check(self.lib.dqmc_udv_decompose_host(self.h, a.ctypes.data, U.ctypes.data, d.ctypes.data_as(_lib._DP),
                                               Vt.ctypes.data, C.byref(sw)))
        return U, d, Vt, sw.value

    def gemm(self, opA, opB, A, B):
        a, b = _fmat(A), _fmat(B)
        c = np.zeros_like(a)
        check(self.lib.dqmc_gemm_host(self.h, opA, opB, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return c

    def setupUdVStorage_and_calculateGreen(self):
        check(self.lib.dqmc_udv_setup(self.h))

    def advanceDownGreen(self, l):
        check(self.lib.dqmc_advance(self.h, DOWN, l))

    def advanceUpGreen(self, l):
        check(self.lib.dqmc_advance(self.h, UP, l))

    def wrapDownGreen(self, k):
        check(self.lib.dqmc_wrap(self.h, DOWN, k))

    def wrapUpGreen(self, k):
        check(self.lib.dqmc_wrap(self.h, UP, k))

    def reset_storage0(self):
        check(self.lib.dqmc_reset_storage0(self.h))

    def push_uniforms(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        check(self.lib.dqmc_push_uniforms_host(self.h, u.ctypes.data_as(_lib._DP), u.size))

    def updateInSlice(self, k, thermalization=False, proposal="box", adapt="box", adaptScaleVariance=False, repeat=1):
        """updateInSlice / updateInSliceThermalization; proposal: box | rotate | scale | rotate_and_scale (the latter three: O(3))"""
        check(self.lib.dqmc_update_slice_ex(self.h, k, int(thermalization), PROPOSE[proposal], ADAPT[adapt], int(adaptScaleVariance), int(repeat)))

    def update_state(self):
        st = _lib.dqmc_update_state()
        check(self.lib.dqmc_get_update_state_host(self.h, C.byref(st)))
        return st

    def set_update_state(self, st):
        check(self.lib.dqmc_set_update_state_host(self.h, C.byref(st)))

    @property
    def g(self):
        g = np.zeros((self.ng, self.ng), dtype=np.complex128, order="F")
        check(self.lib.dqmc_get_green_host(self.h, g.ctypes.data))
        return g

    def set_green(self, g, k):
        a = _fmat(g)
        check(self.lib.dqmc_set_green_host(self.h, a.ctypes.data, k))

    @property
    def g_inv_sv(self):
        sv = np.zeros(self.ng)
        check(self.lib.dqmc_get_sv_host(self.h, sv.ctypes.data_as(_lib._DP)))
        return sv

    def udv(self, l):
        U = np.zeros((self.ng, self.ng), dtype=np.complex128, order="F")
        Vt = np.zeros_like(U)
        d = np.zeros(self.ng)
        check(self.lib.dqmc_get_udv_host(self.h, l, U.ctypes.data, d.ctypes.data_as(_lib._DP), Vt.ctypes.data))
        return U, d, Vt

    @property
    def currentTimeslice(self):
        return self.lib.dqmc_current_timeslice(self.h)

    def backup(self):
        check(self.lib.dqmc_backup(self.h))

    def restore(self):
        check(self.lib.dqmc_restore(self.h))

    def exchange_action(self):
        v = C.c_double(0)
        check(self.lib.dqmc_exchange_action_host(self.h, C.byref(v)))
        return v.value

    def set_exchange_parameter(self, r):
        check(self.lib.dqmc_set_exchange_parameter(self.h, r))

    # one transfer / one launch for ALL chains of a batched context
    def phi_action_all(self):
        out = np.zeros(self.nchains_total())
        check(self.lib.dqmc_phi_action_all_host(self.h, out.ctypes.data_as(_lib._DP)))
        return out

    def shift_fields_all(self, shifts):
        a = np.ascontiguousarray(shifts, dtype=np.float64)
        check(self.lib.dqmc_shift_fields_all_host(self.h, a.ctypes.data_as(_lib._DP)))

    def get_fields_all(self):
        """(nchains, m+1, N, OPDIM)"""
        nb = self.nchains_total()
        a = np.zeros((nb, self.m + 1, self.opdim, self.N))
        check(self.lib.dqmc_get_fields_all_host(self.h, a.ctypes.data_as(_lib._DP)))
        return np.transpose(a, (0, 1, 3, 2)).copy()

    def sv_all(self):
        a = np.zeros((self.nchains_total(), self.ng))
        check(self.lib.dqmc_get_sv_all_host(self.h, a.ctypes.data_as(_lib._DP)))
        return a

    def nchains_total(self):
        return self.lib.dqmc_num_chains(self.h)

    def synchronize(self):
        check(self.lib.dqmc_synchronize(self.h))

    def schedule_info(self):
        """which execution variants this context latched at create time, and how many update blocks ran through each schedule"""
        si = _lib.dqmc_schedule_info()
        check(self.lib.dqmc_get_schedule_info(self.h, C.byref(si)))
        return si

    def profile_enable(self, on=True):
        check(self.lib.dqmc_profile_enable(self.h, int(on)))

    def profile_read(self):
        pr = _lib.dqmc_profile()
        check(self.lib.dqmc_profile_read(self.h, C.byref(pr)))
        names = ["bmult", "gemm", "decomp", "decide", "other", "gather", "flush"]
        out = {nm: (pr.ms[i], int(pr.launches[i])) for i, nm in enumerate(names)}
        out["jacobi"] = out["decomp"]
        out.update(svd_calls=int(pr.svd_calls), svd_sweeps_total=int(pr.svd_sweeps_total),
                   svd_sweeps_max=int(pr.svd_sweeps_max), qr_calls=int(pr.qr_calls), gemm_flops=pr.gemm_flops,
                   decomp_round_ms=pr.decomp_round_ms, decomp_rounds=int(pr.decomp_rounds),
                   blocks_nonempty=int(pr.blocks_nonempty), chains=int(pr.chains),
                   updates_accepted=int(pr.updates_accepted), lu_calls=int(pr.lu_calls),
                   sub={"lu_update": [pr.sub_ms[0], int(pr.sub_launches[0]), pr.sub_flops[0], pr.sub_bytes[0]],
                        "fact_gemm": [pr.sub_ms[1], int(pr.sub_launches[1]), pr.sub_flops[1], pr.sub_bytes[1]]})
        return out


class _CtxView(KernelContext):
    """Non-owning KernelContext over the dqmc_ctx of a DetSDW replica (profiling, state inspection)."""

    def __init__(self, lib, handle, info):
        self.lib, self.h = lib, C.c_void_p(handle)
        self.opdim, self.L, self.m, self.s = info.opdim, info.L, info.m, info.s
        self.N, self.MSF, self.ng, self.n = info.N, info.MSF, info.n_g, info.n

    def close(self):
        self.h = None


def _host_params(pars: SDWParams):
    return _lib.detsdw_params(
        opdim=pars.opdim, L=pars.L, m=pars.m, s=pars.s, delaySteps=pars.delaySteps,
        globalShift=int(pars.globalShift), globalUpdateInterval=pars.globalUpdateInterval,
        weakZflux=int(pars.weakZflux), phi2bosons=int(pars.phi2bosons), device=pars.device,
        simindex=pars.simindex, rngSeed=pars.rngSeed,
        has_mux_muy=int(pars.mux is not None and pars.muy is not None),
        updateMethod=UPDATE_METHOD[pars.updateMethod], bc=pars.bc.encode(),
        beta=pars.beta, dtau=pars.dtau, r=pars.r, c=pars.c, u=pars.u, lambda_=pars.lambda_,
        txhor=pars.txhor, txver=pars.txver, tyhor=pars.tyhor, tyver=pars.tyver,
        mu=pars.mu, mux=pars.mux or 0.0, muy=pars.muy or 0.0, accRatio=pars.accRatio, cdwU=pars.cdwU,
        stabilisation=STABILISATION[pars.stabilisation], cb_none=int(not pars.checkerboard),
        wolffClusterUpdate=int(pars.wolffClusterUpdate), wolffClusterShiftUpdate=int(pars.wolffClusterShiftUpdate),
        repeatWolffPerSweep=int(pars.repeatWolffPerSweep), fermionMeasurements=int(pars.fermionMeasurements),
        spinProposalMethod=SPIN_PROPOSAL[pars.spinProposalMethod], adaptScaleVariance=int(pars.adaptScaleVariance),
        repeatUpdateInSlice=int(pars.repeatUpdateInSlice),
        tuning=_tuning(pars.pipeline, pars.qrVariant, pars.greenVariant, pars.maxJacobiSweeps, pars.proposalBudget, pars.decideThreads))


class DetSDW:
    """The replica (C++ host layer): same method names as the reference's DetSDW / DetModel.

    Also serves as the view of ONE chain of a DetSDWBatch (`batch.chain(b)`): then it does not own the
    handle, `sweep*` are not available on it (the batch sweeps all chains in lockstep) and every call first
    selects its chain."""

    def __init__(self, pars: SDWParams = None, _batch=None, _chain=0):
        self.lib = load()
        self._chain = _chain
        self._batch = _batch
        if _batch is not None:
            self.h = _batch.h
            self.pars = _batch.pars_list[_chain]
            return
        p = _host_params(pars)
        h = C.c_void_p()
        check(self.lib.detsdw_create(C.byref(p), C.byref(h)), host=True)
        self.h = h
        self.pars = pars

    def _sel(self):
        if self._batch is not None:
            check(self.lib.detsdw_select_chain(self.h, self._chain), host=True)

    def close(self):
        if self._batch is not None:
            self.h = None
            return
        if self.h:
            self.lib.detsdw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, takeMeasurements=False):
        if self._batch is not None:
            raise RuntimeError("a chain of a batch cannot sweep on its own: call DetSDWBatch.sweep()")
        check(self.lib.detsdw_sweep(self.h, int(takeMeasurements)), host=True)

    def sweepThermalization(self):
        if self._batch is not None:
            raise RuntimeError("a chain of a batch cannot sweep on its own: call DetSDWBatch.sweepThermalization()")
        check(self.lib.detsdw_sweep_thermalization(self.h), host=True)

    @property
    def info(self):
        self._sel()
        i = _lib.detsdw_info()
        check(self.lib.detsdw_get_info(self.h, C.byref(i)), host=True)
        return i

    @property
    def observables(self):
        """bosonic observables of the last sweep(takeMeasurements=True) (reference: measure / finishMeasurements)"""
        self._sel()
        o = _lib.detsdw_observables()
        check(self.lib.detsdw_get_observables(self.h, C.byref(o)), host=True)
        return o

    def observable_vector(self, name):
        """'kOccX', 'kOccY', 'pairPlus', 'pairMinus' of the last sweep(True) with fermionMeasurements"""
        self._sel()
        out = np.zeros(self.info.N)
        which = {"kOccX": 0, "kOccY": 1, "pairPlus": 2, "pairMinus": 3}[name]
        check(self.lib.detsdw_get_observable_vector(self.h, which, out.ctypes.data_as(_lib._DP)), host=True)
        return out

    @property
    def phi(self):
        """(m+1, N, OPDIM) like the oracle; the ABI hands out the reference layout."""
        self._sel()
        i = self.info
        a = np.zeros((i.N, i.opdim, i.m + 1), order="F")
        check(self.lib.detsdw_get_phi(self.h, a.ctypes.data_as(_lib._DP)), host=True)
        return np.transpose(a, (2, 0, 1)).copy()

    @property
    def cdwl(self):
        """(m+1, N) int32: the discrete field l_i(tau_k) of cdwU != 0 (drawn at set-up whatever cdwU is; slice 0 unused)."""
        self._sel()
        i = self.info
        a = np.zeros((i.m + 1, i.N), dtype=np.int32)
        check(self.lib.detsdw_get_cdwl(self.h, a.ctypes.data), host=True)
        return a

    def set_cdwl(self, cdwl_kN):
        self._sel()
        a = np.ascontiguousarray(cdwl_kN, dtype=np.int32)
        check(self.lib.detsdw_set_cdwl(self.h, a.ctypes.data), host=True)

    def set_phi(self, phi_kNd):
        self._sel()
        ref = np.asfortranarray(np.transpose(np.asarray(phi_kNd, dtype=np.float64), (1, 2, 0)))
        check(self.lib.detsdw_set_phi(self.h, ref.ctypes.data_as(_lib._DP)), host=True)

    @property
    def g(self):
        self._sel()
        n = self.info.n_g
        g = np.zeros((n, n), dtype=np.complex128, order="F")
        check(self.lib.detsdw_get_green(self.h, g.ctypes.data), host=True)
        return g

    @property
    def g_inv_sv(self):
        self._sel()
        sv = np.zeros(self.info.n_g)
        check(self.lib.detsdw_get_green_inv_sv(self.h, sv.ctypes.data_as(_lib._DP)), host=True)
        return sv

    def save_state(self, path):
        """checkpoint of the whole replica object (all chains if this is a view of a batch)"""
        check(self.lib.detsdw_save_state(self.h, str(path).encode()), host=True)

    def load_state(self, path):
        check(self.lib.detsdw_load_state(self.h, str(path).encode()), host=True)

    def saveConfigurationStreamBinary(self, directory="."):
        """appends to <directory>/configs-phi.binarystream (reference format, src/detsdwopdim.cpp:4991-5012)"""
        self._sel()
        check(self.lib.detsdw_save_configuration_stream_binary(self.h, str(directory).encode()), host=True)

    def rand01(self):
        self._sel()
        return self.lib.detsdw_rng_rand01(self.h)

    @property
    def kernel_context(self):
        """the kernel context that holds this chain (a batch may spread its chains over several), this chain selected"""
        local = C.c_int(0)
        kc = _CtxView(self.lib, self.lib.detsdw_ctx_of_chain(self.h, self._chain, C.byref(local)), self.info)
        check(self.lib.dqmc_select_chain(kc.h, local.value))
        return kc

    # replica exchange surface (reference src/detsdwopdim.h:116-153)
    def get_exchange_parameter_value(self):
        self._sel()
        return self.lib.detsdw_get_exchange_parameter_value(self.h)

    def set_exchange_parameter_value(self, v):
        self._sel()
        check(self.lib.detsdw_set_exchange_parameter_value(self.h, v), host=True)

    def get_exchange_parameter_name(self):
        return self.lib.detsdw_get_exchange_parameter_name(self.h).decode()

    def get_exchange_action_contribution(self):
        self._sel()
        v = C.c_double(0)
        check(self.lib.detsdw_get_exchange_action_contribution(self.h, C.byref(v)), host=True)
        return v.value

    def get_control_data(self):
        self._sel()
        cd = _lib.detsdw_control_data()
        check(self.lib.detsdw_get_control_data(self.h, C.byref(cd)), host=True)
        return cd

    def set_control_data(self, cd):
        self._sel()
        check(self.lib.detsdw_set_control_data(self.h, C.byref(cd)), host=True)


class DetSDWBatch:
    """The replicas of one parallel-tempering ensemble on ONE GPU, swept in lockstep by one kernel context
    (detsdw_create_batch): every launch carries all chains, which is what fills the chip.  The parameter sets
    may differ only in r, rngSeed and simindex; chain b follows exactly the Markov chain of DetSDW(pars[b])."""

    def __init__(self, pars_list, sub_batches=0):
        """sub_batches: kernel contexts the chains are spread over and swept concurrently (one host thread each);
        0 = automatic (up to 4 contexts of at least 32 chains), 1 = one context / one launch sequence for all chains"""
        self.lib = load()
        self.pars_list = list(pars_list)
        arr = (_lib.detsdw_params * len(self.pars_list))(*[_host_params(p) for p in self.pars_list])
        h = C.c_void_p()
        check(self.lib.detsdw_create_batch_ex(arr, len(self.pars_list), int(sub_batches), C.byref(h)), host=True)
        self.h = h
        self.sub_batches = self.lib.detsdw_num_sub_batches(h)
        self.chains = [DetSDW(_batch=self, _chain=b) for b in range(len(self.pars_list))]

    def __len__(self):
        return len(self.chains)

    def chain(self, b):
        return self.chains[b]

    def sweep(self, takeMeasurements=False):
        check(self.lib.detsdw_sweep(self.h, int(takeMeasurements)), host=True)

    def sweepThermalization(self):
        check(self.lib.detsdw_sweep_thermalization(self.h), host=True)

    def save_state(self, path):
        check(self.lib.detsdw_save_state(self.h, str(path).encode()), host=True)

    def load_state(self, path):
        check(self.lib.detsdw_load_state(self.h, str(path).encode()), host=True)

    def exchange_actions_device(self, device_ptr):
        """get_exchange_action_contribution of EVERY chain written to device memory (len(self) doubles at device_ptr, e.g.
        torch_tensor.data_ptr()): the send buffer of the replica-exchange all_gather, no host hop"""
        check(self.lib.detsdw_exchange_actions_device(self.h, C.c_void_p(int(device_ptr))), host=True)

    @property
    def kernel_context(self):
        """the kernel context of chain 0 (the only one unless the batch has several sub-batches)"""
        return self.chains[0].kernel_context

    def kernel_contexts(self):
        """one view per sub-batch"""
        per = len(self.chains) // self.sub_batches
        return [self.chains[g * per].kernel_context for g in range(self.sub_batches)]

    def close(self):
        if self.h:
            for c in self.chains:
                c.h = None
            self.lib.detsdw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class HubbardParams:
    """ModelParams<DetHubbard> (reference src/dethubbardparams.h:28-50)"""
    L: int = 4
    d: int = 2
    beta: float = 0.0
    m: int = 0
    dtau: float = 0.1
    s: int = 10
    t: float = 1.0
    U: float = 4.0
    mu: float = 0.0
    checkerboard: bool = False
    rngSeed: int = 1020304050
    simindex: int = 0
    device: int = 0
    stabilisation: str = "svd"


class DetHubbard:
    """The Hubbard replica of BASELINE config 1 (C++ host layer detqmc_amd/csrc/host/dethubbard.cpp): the reference's
    DetHubbard (src/dethubbard.{h,cpp}) with both spin sectors in one block-diagonal Green's function on the GPU.
    nchains > 1: independent replicas (simindex, simindex + 1, ...) in lockstep; `select(b)` picks the one the getters
    talk to."""

    def __init__(self, pars: HubbardParams, nchains=1):
        self.lib = load()
        self.pars = pars
        p = _lib.dethubbard_params(L=pars.L, d=pars.d, m=pars.m, s=pars.s, checkerboard=int(pars.checkerboard), device=pars.device,
                                   simindex=pars.simindex, rngSeed=pars.rngSeed, stabilisation=STABILISATION[pars.stabilisation],
                                   beta=pars.beta, dtau=pars.dtau, t=pars.t, U=pars.U, mu=pars.mu)
        h = C.c_void_p()
        check(self.lib.dethubbard_create(C.byref(p), nchains, C.byref(h)), host="hubbard")
        self.h = h
        self.nchains = nchains

    def close(self):
        if self.h:
            self.lib.dethubbard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def select(self, b):
        check(self.lib.dethubbard_select_chain(self.h, b), host="hubbard")

    def sweep(self, takeMeasurements=False):
        check(self.lib.dethubbard_sweep(self.h, int(takeMeasurements)), host="hubbard")

    def sweepThermalization(self):
        check(self.lib.dethubbard_sweep_thermalization(self.h), host="hubbard")

    @property
    def info(self):
        i = _lib.dethubbard_info()
        check(self.lib.dethubbard_get_info(self.h, C.byref(i)), host="hubbard")
        return i

    @property
    def auxfield(self):
        """(N, m+1) like the reference's MatInt auxfield, as +-1.0 (column 0 unused)"""
        i = self.info
        a = np.zeros((i.N, i.m + 1), order="F")
        check(self.lib.dethubbard_get_auxfield(self.h, a.ctypes.data_as(_lib._DP)), host="hubbard")
        return a

    @property
    def green(self):
        """(gUp, gDn), N x N each"""
        n = self.info.N
        gu, gd = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        check(self.lib.dethubbard_get_green(self.h, gu.ctypes.data_as(_lib._DP), gd.ctypes.data_as(_lib._DP)), host="hubbard")
        return gu, gd

    @property
    def observables(self):
        o = _lib.dethubbard_observables()
        check(self.lib.dethubbard_get_observables(self.h, C.byref(o)), host="hubbard")
        return o

    @property
    def zcorr(self):
        out = np.zeros(self.info.N)
        check(self.lib.dethubbard_get_zcorr(self.h, out.ctypes.data_as(_lib._DP)), host="hubbard")
        return out

    def rand01(self):
        return self.lib.dethubbard_rng_rand01(self.h)

    def save_state(self, path):
        check(self.lib.dethubbard_save_state(self.h, str(path).encode()), host="hubbard")

    def load_state(self, path):
        check(self.lib.dethubbard_load_state(self.h, str(path).encode()), host="hubbard")
