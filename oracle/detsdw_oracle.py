"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product package `detqmc_amd`;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

CPU restatement (numpy + scipy/LAPACK zgesvd) of the reference's DetModelGC/DetSDW sweep hot
path, following the reference algorithm literally (SVD-based UdV, checkerboard B-multiplies,
delayed rank-MSF*D updates in the X/Y form).  Every function cites the reference file:line it
follows; paths are relative to /root/reference/src.

Pinned against the REAL reference: tests/golden/*.npz are produced by oracle/make_golden.py
from oracle/_ref/ref_harness_o{1,2,3} (the reference sources compiled where they lie, see
oracle/ref_build/).  tests/test_oracle_vs_golden.py checks this file against them.

Conventions: matrices are numpy [row, col]; site = y*L + x (neighbortable.h:53-80); block index
b of an n_g = MSF*N matrix selects rows/cols b*N..(b+1)*N-1; bands: even block -> XBAND,
odd block -> YBAND (detsdwopdim.cpp:2045-2065).
"""
import math
from dataclasses import dataclass, field
import numpy as np
import scipy.linalg as sla

from dsfmt_oracle import RngWrapper

XBAND, YBAND = 0, 1


# ---------------------------------------------------------------------------------------------
# parameters: detsdwparams.h:24-120, detsdwparams.cpp:21-140, detmodelparams.h:68-122
# ---------------------------------------------------------------------------------------------
@dataclass
class SDWParams:
    opdim: int = 2
    L: int = 4
    beta: float = 2.0
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
    cdwU: float = 0.0             # != 0: discrete four-valued field l_i(tau) coupled to the band-charge difference (detsdwparams.h:61)
    bc: str = "pbc"
    weakZflux: bool = False
    globalShift: bool = False
    wolffClusterUpdate: bool = False
    wolffClusterShiftUpdate: bool = False
    repeatWolffPerSweep: int = 1
    globalUpdateInterval: int = 100
    phi2bosons: bool = False
    turnoffFermionMeasurements: bool = True   # False: measure() also takes the G-dependent observables
    checkerboard: bool = True     # False = CB_NONE: dense B = e^{-dtau V} e^{-dtau K} (detsdwopdim.h:1305-1375)
    spinProposalMethod: str = "box"           # "box", "rotate_then_scale", "rotate_and_scale" (the latter two: opdim 3 only; detsdwparams.h:40)
    adaptScaleVariance: bool = False          # detsdwparams.h:43
    repeatUpdateInSlice: int = 1              # detsdwparams.h:90
    rngSeed: int = 1020304050
    simindex: int = 0
    # derived
    m: int = 0
    n: int = 0
    N: int = 0

    def finalize(self):
        # updateTemperatureParameters (detmodelparams.h:68-122)
        self.m = int(round(self.beta / self.dtau))
        self.beta = self.m * self.dtau
        while self.m <= self.s:
            self.s -= 1
        if self.s < 1:
            raise ValueError("Cannot choose parameter s obeying 0 < s < m")
        # ModelParamsDetSDW::check (detsdwparams.cpp:21-140)
        if self.opdim not in (1, 2, 3):
            raise ValueError("opdim")
        if self.bc not in ("pbc", "apbc-x", "apbc-y", "apbc-xy"):
            raise ValueError("bc")
        if self.weakZflux and self.opdim != 2:
            raise ValueError("Magnetic field only supported for opdim=2")
        if self.L % 2 != 0:
            raise ValueError("Checker board decomposition only supported for even linear lattice sizes")
        self.N = self.L * self.L
        if self.delaySteps <= 0 or self.delaySteps > self.N:
            raise ValueError("delaySteps")
        if (self.globalShift or self.wolffClusterUpdate or self.wolffClusterShiftUpdate) and self.globalUpdateInterval == 0:
            raise ValueError("globalUpdateInterval")            # detsdwparams.cpp:89-93
        if self.wolffClusterShiftUpdate and (self.globalShift or self.wolffClusterUpdate):
            raise ValueError("Either use combined wolffClusterShiftUpdate or individual global updates")   # :94-96
        if self.repeatWolffPerSweep < 1:
            raise ValueError("repeatWolffPerSweep")
        # createReplica (detsdwopdim.cpp:75-79)
        if self.mux is None or self.muy is None:
            self.mux = self.mu
            self.muy = self.mu
        # DetModelGC ctor (detmodel.h:518)
        self.n = int(math.ceil(self.m / self.s))
        return self


class RunningAverage:
    """RunningAverage.h:19-80."""

    def __init__(self, sampleSize):
        self.sampleSize = sampleSize
        self.samplesAdded = 0
        self.values = []
        self.runningAverage = 0.0

    def addValue(self, v):
        if self.samplesAdded < self.sampleSize:
            self.values.append(v)
            self.runningAverage += v / self.sampleSize
        else:
            self.runningAverage -= self.values[0] / self.sampleSize
            self.values.pop(0)
            self.values.append(v)
            self.runningAverage += v / self.sampleSize
        self.samplesAdded += 1

    def get(self):
        return self.runningAverage


def _libm_sincos():
    """The reference, built with g++ -O2, evaluates cos(phi) and sin(phi) of proposeRandomRotatedVector through ONE glibc sincos()
    call, whose results differ from sin() / cos() in the last bit for ~0.1 % of the arguments (measured in this image, glibc 2.35):
    the oracle calls the same function so that rotated fields stay bit-identical to the reference fixtures."""
    import ctypes
    try:
        libm = ctypes.CDLL("libm.so.6")
        libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        libm.sincos.restype = None
    except (OSError, AttributeError):
        return lambda x: (math.sin(x), math.cos(x))
    sv, cv = ctypes.c_double(), ctypes.c_double()

    def f(x):
        libm.sincos(x, ctypes.byref(sv), ctypes.byref(cv))
        return sv.value, cv.value
    return f


_sincos = _libm_sincos()


class NormalDistribution:
    """normaldistribution.h:25-80: polar Box-Muller on the replica's RngWrapper, the two values of a pair on a stack (the
    second one generated is handed out first); reset() at the top of every updateInSlice (detsdwopdim.cpp:2433-2435)."""

    def __init__(self, rng):
        self.rng = rng
        self.stack = []

    def reset(self):
        self.stack = []

    def get(self, sigma, mean):
        if not self.stack:
            while True:
                u1 = self.rng.rand01()
                u2 = self.rng.rand01()
                v1 = 2.0 * u1 - 1.0
                v2 = 2.0 * u2 - 1.0
                rsq = v1 * v1 + v2 * v2
                if not (rsq >= 1.0 or rsq == 0.0):
                    break
            fac = math.sqrt(-2.0 * math.log(rsq) / rsq)
            self.stack.append(v1 * fac)
            self.stack.append(v2 * fac)
        var = self.stack.pop()
        return mean + sigma * var


class UdV:
    """udv.h:40-65.  M = U diag(d) V_t^dagger."""

    def __init__(self, U, d, V_t):
        self.U, self.d, self.V_t = U, d, V_t

    @staticmethod
    def eye(n):
        return UdV(np.eye(n, dtype=complex), np.ones(n), np.eye(n, dtype=complex))

    def copy(self):
        return UdV(self.U.copy(), self.d.copy(), self.V_t.copy())


def udvDecompose(M):
    """udv.h:68-90: arma::svd(U, d, V_t, M, "std") -> LAPACK zgesvd, M = U diag(d) V_t^H."""
    U, d, Vh = sla.svd(M, full_matrices=True, lapack_driver="gesvd", check_finite=False)
    return UdV(U, d, Vh.conj().T)


def make_test_matrix(n):
    """Deterministic asymmetric test matrix shared with oracle/ref_build/ref_harness.cpp."""
    i = np.arange(n, dtype=float)[:, None]
    j = np.arange(n, dtype=float)[None, :]
    return np.sin(0.37 * i + 1.31 * j + 0.11 * i * j) + 1j * np.cos(0.73 * i - 0.29 * j + 0.05 * i * j)


def replica_exchange_probability(par1, action1, par2, action2):
    """detsdwopdim.cpp:5251-5264."""
    delta = (par1 - par2) * (action2 - action1)
    return 1.0 if delta <= 0.0 else math.exp(-delta)


class DetSDWOracle:
    # AdjustmentData constants, detsdwopdim.h:489-498
    InitialPhiDelta = 0.5
    InitialAngleDelta = 0.0
    InitialScaleDelta = 0.1
    MinScaleDelta, MaxScaleDelta = 0.0, 1.0
    MinAngleDelta, MaxAngleDelta = -1.0, 1.0
    AccRatioAdjustmentSamples = 100
    phiDeltaGrowFactor = 1.05
    phiDeltaShrinkFactor = 0.95

    def __init__(self, pars: SDWParams, rng: RngWrapper = None, phi=None, cdwl=None):
        """DetSDW ctor, detsdwopdim.cpp:158-361."""
        p = pars.finalize() if pars.m == 0 else pars
        self.pars = p
        self.rng = rng if rng is not None else RngWrapper(p.rngSeed, p.simindex + 1)  # detqmc.h:181
        self.OPDIM = p.opdim
        self.MSF = 4 if p.opdim == 3 else 2                                           # detsdwopdim.h:161
        self.N, self.L, self.m, self.s, self.n = p.N, p.L, p.m, p.s, p.n
        self.ng = self.MSF * self.N
        self.dtau = p.dtau
        N, m = self.N, self.m
        self.phi = np.zeros((m + 1, N, self.OPDIM))
        self.coshTermPhi = np.zeros((m + 1, N))
        self.sinhTermPhi = np.zeros((m + 1, N))
        # discrete field and its caches (ctor :176-209: cdwl.zeros(), coshTermCDWl.ones(), sinhTermCDWl.zeros())
        self.cdwl = np.zeros((m + 1, N), dtype=np.int32)
        self.coshTermCDWl = np.ones((m + 1, N))
        self.sinhTermCDWl = np.zeros((m + 1, N))
        # AdjustmentData (detsdwopdim.h:481-577) and UpdateStatistics (:285-311)
        self.phiDelta = self.InitialPhiDelta
        self.targetAccRatioLocal_phi = p.accRatio
        self.lastAccRatioLocal_phi = 0.0
        self.accRatioLocal_box_RA = RunningAverage(self.AccRatioAdjustmentSamples)
        # rotate / scale proposals (detsdwopdim.h:489-530)
        self.angleDelta = self.InitialAngleDelta
        self.scaleDelta = self.InitialScaleDelta
        self.accRatioLocal_rotate_RA = RunningAverage(self.AccRatioAdjustmentSamples)
        self.accRatioLocal_scale_RA = RunningAverage(self.AccRatioAdjustmentSamples)
        self.curminAngleDelta, self.curmaxAngleDelta = self.MinAngleDelta, self.MaxAngleDelta
        self.curminScaleDelta, self.curmaxScaleDelta = self.MinScaleDelta, self.MaxScaleDelta
        self.normal_distribution = NormalDistribution(self.rng)
        if p.spinProposalMethod != "box" and p.opdim != 3:
            raise ValueError("rotate / scale proposals are only supported for the O(3) model (detsdwopdim.cpp:3934-3939)")
        self.acceptedGlobalShifts = 0
        self.attemptedGlobalShifts = 0
        self.acceptedWolffClusterUpdates = 0
        self.attemptedWolffClusterUpdates = 0
        self.acceptedWolffClusterShiftUpdates = 0
        self.attemptedWolffClusterShiftUpdates = 0
        self.addedWolffClusterSize = 0.0
        self.performedSweeps = 0
        self._setup_lattice()
        if phi is None:
            self.setupRandomField()
        else:
            self.phi[:] = phi
            self.cdwl[1:] = 1 if cdwl is None else np.asarray(cdwl)[1:]
            self.updateCoshSinhTermsPhi()
            if p.cdwU:
                self.updateCoshSinhTermsCDWl()
        self._setup_hopping()
        self.g = np.zeros((self.ng, self.ng), dtype=complex)
        self.g_inv_sv = np.zeros(self.ng)
        self.UdVStorage = None
        self.currentTimeslice = 0
        self.lastSweepDir = +1  # Up
        self.setupUdVStorage_and_calculateGreen()

    # ------------------------------------------------------------------ lattice / fields
    def _setup_lattice(self):
        """neighbortable.h:53-80: site = y*L + x; dirs XPLUS,XMINUS,YPLUS,YMINUS."""
        L = self.L
        x = np.arange(self.N) % L
        y = np.arange(self.N) // L
        self.neigh = np.stack([
            y * L + (x + 1) % L,
            y * L + (x - 1 + L) % L,
            ((y + 1) % L) * L + x,
            ((y - 1 + L) % L) * L + x,
        ])  # [dir, site]

    @staticmethod
    def cdwl_from_uniform(r):
        """The four-way draw of setupRandomField (:1105-1109) and proposeNewCDWl (:4176-4180)."""
        if r <= 0.25:
            return +2
        if r <= 0.5:
            return -2
        if r <= 0.75:
            return +1
        return -1

    @staticmethod
    def cdwl_gamma(l):
        """detsdwopdim.h:1209-1220."""
        return (3.0 + math.sqrt(6.0)) if abs(l) == 1 else ((3.0 - math.sqrt(6.0)) if abs(l) == 2 else 0.0)

    @staticmethod
    def cdwl_eta(l):
        """detsdwopdim.h:1222-1235."""
        if abs(l) == 1:
            return math.copysign(math.sqrt(2.0 * (3.0 - math.sqrt(6.0))), l)
        if abs(l) == 2:
            return math.copysign(math.sqrt(2.0 * (3.0 + math.sqrt(6.0))), l)
        return 0.0

    def getCoshSinhTermCDWl(self, l):
        """detsdwopdim.cpp:1138-1143."""
        arg = math.sqrt(self.dtau) * self.pars.cdwU * self.cdwl_eta(int(l))
        return math.cosh(arg), math.sinh(arg)

    def updateCoshSinhTermsCDWl(self):
        """detsdwopdim.cpp:1183-1190 (only called when cdwU != 0, :1148-1151)."""
        for k in range(1, self.m + 1):
            for site in range(self.N):
                self.coshTermCDWl[k, site], self.sinhTermCDWl[k, site] = self.getCoshSinhTermCDWl(self.cdwl[k, site])

    def setupRandomField(self):
        """detsdwopdim.cpp:1099-1113: k outer, site, dim; one more rand01 per site sets cdwl (whatever cdwU is)."""
        rng = self.rng
        for k in range(1, self.m + 1):
            for site in range(self.N):
                for dim in range(self.OPDIM):
                    self.phi[k, site, dim] = rng.randRange(-1.0, 1.0)
                self.cdwl[k, site] = self.cdwl_from_uniform(rng.rand01())
        self.updateCoshSinhTermsPhi()
        if self.pars.cdwU:
            self.updateCoshSinhTermsCDWl()

    def getCoshSinhTermPhi(self, phivec):
        """detsdwopdim.cpp:1132-1136."""
        nrm = math.sqrt(float(np.dot(phivec, phivec)))
        a = self.pars.lambda_ * self.dtau * nrm
        return math.cosh(a), math.sinh(a) / nrm

    def updateCoshSinhTermsPhi(self):
        """detsdwopdim.cpp:1175-1181."""
        nrm = np.sqrt(np.sum(self.phi[1:] ** 2, axis=2))
        a = self.pars.lambda_ * self.dtau * nrm
        self.coshTermPhi[1:] = np.cosh(a)
        self.sinhTermPhi[1:] = np.sinh(a) / nrm

    # ------------------------------------------------------------------ hopping / checkerboard
    def _setup_hopping(self):
        """detsdwopdim.cpp:217-260 (constants), :1598-1684 (flux 4-site exponentials),
        :1788-1826 (no-flux plaquette factors).  Builds, per (band, kind), the list of
        plaquette site quadruples [i,j,k,l] per subgroup and their 4x4 matrices.
        kind: ('full'|'half', sign)."""
        p, L, dtau = self.pars, self.L, self.dtau
        hopHor = {XBAND: p.txhor, YBAND: p.tyhor}
        hopVer = {XBAND: p.txver, YBAND: p.tyver}
        self.mu_band = {XBAND: p.mux, YBAND: p.muy}
        zmag = {XBAND: (1.0 / self.N if p.weakZflux else 0.0),       # zmag[XUP]
                YBAND: (1.0 / self.N if p.weakZflux else 0.0)}       # zmag[YDOWN]
        apbc_x = p.bc in ("apbc-x", "apbc-xy")
        apbc_y = p.bc in ("apbc-y", "apbc-xy")
        self.plaq_sites = {}
        self.plaq_mats = {}
        for subgroup in (0, 1):
            quads = []
            for i1 in range(subgroup, L, 2):          # x
                for i2 in range(subgroup, L, 2):      # y
                    i = i2 * L + i1
                    j = self.neigh[0, i]
                    k = self.neigh[2, i]
                    l = self.neigh[0, k]
                    quads.append((i, j, k, l, i1, i2))
            self.plaq_sites[subgroup] = np.array([q[:4] for q in quads], dtype=np.int64)
            for band in (XBAND, YBAND):
                for half in (False, True):
                    for sign in (-1, +1):
                        mats = np.zeros((len(quads), 4, 4), dtype=complex)
                        for idx, (i, j, k, l, i1, i2) in enumerate(quads):
                            if not p.weakZflux:
                                f = 0.5 if half else 1.0
                                ch_hor = math.cosh(-f * dtau * hopHor[band])
                                sh_hor = sign * math.sinh(-f * dtau * hopHor[band])
                                ch_ver = math.cosh(-f * dtau * hopVer[band])
                                sh_ver = sign * math.sinh(-f * dtau * hopVer[band])
                                if apbc_x and i1 == L - 1:
                                    sh_hor *= -1
                                if apbc_y and i2 == L - 1:
                                    sh_ver *= -1
                                a = ch_hor * ch_ver
                                b = ch_ver * sh_hor
                                c = ch_hor * sh_ver
                                d = sh_hor * sh_ver
                                mats[idx] = [[a, b, c, d], [b, a, d, c], [c, d, a, b], [d, c, b, a]]
                            else:
                                hh, hv = hopHor[band], hopVer[band]
                                if apbc_x and i1 == L - 1:
                                    hh *= -1
                                if apbc_y and i2 == L - 1:
                                    hv *= -1
                                zm = zmag[band]
                                j1 = j % L
                                k2 = k // L
                                ph_ij = np.exp(1j * (-2.0 * math.pi * zm * i2))
                                ph_kl = np.exp(1j * (-2.0 * math.pi * zm * k2))
                                ph_ik = 1.0 + 0j
                                ph_jl = 1.0 + 0j
                                if i2 == L - 1:
                                    ph_ik = np.exp(1j * (2.0 * math.pi * zm * L * i1))
                                    ph_jl = np.exp(1j * (2.0 * math.pi * zm * L * j1))
                                H = np.zeros((4, 4), dtype=complex)
                                H[0, 1] = ph_ij * hh
                                H[0, 2] = ph_ik * hv
                                H[1, 3] = ph_jl * hv
                                H[2, 3] = ph_kl * hh
                                H = -(H + H.conj().T)
                                ev, evec = np.linalg.eigh(H)
                                pref = sign * (0.5 if half else 1.0) * dtau
                                mats[idx] = (evec * np.exp(pref * ev)) @ evec.conj().T
                        self.plaq_mats[(band, subgroup, half, sign)] = mats

    def _apply_plaq_left(self, R, subgroup, mats):
        """rows[i,j,k,l] <- mat . rows   (detsdwopdim.cpp:1688-1720, :1788-1826)."""
        q = self.plaq_sites[subgroup]
        rows = R[q]                                   # [P, 4, ncols]
        R[q] = np.einsum("pab,pbc->pac", mats, rows)

    def _apply_plaq_right(self, R, subgroup, mats):
        """cols[i,j,k,l] <- cols . mat   (detsdwopdim.cpp:1722-1756, :1905-1943)."""
        q = self.plaq_sites[subgroup]
        cols = R[:, q]                                # [nrows, P, 4]
        R[:, q] = np.einsum("rpb,pba->rpa", cols, mats)

    def cbLMultHoppingExp(self, A, band, sign):
        """detsdwopdim.cpp:1839-1869: e^{sign dtau K1/2} e^{sign dtau K0} e^{sign dtau K1/2} A."""
        R = np.array(A, dtype=complex, copy=True)
        self._apply_plaq_left(R, 1, self.plaq_mats[(band, 1, True, sign)])
        self._apply_plaq_left(R, 0, self.plaq_mats[(band, 0, False, sign)])
        self._apply_plaq_left(R, 1, self.plaq_mats[(band, 1, True, sign)])
        return R

    def cbRMultHoppingExp(self, A, band, sign):
        """detsdwopdim.cpp:1948-1979."""
        R = np.array(A, dtype=complex, copy=True)
        self._apply_plaq_right(R, 1, self.plaq_mats[(band, 1, True, sign)])
        self._apply_plaq_right(R, 0, self.plaq_mats[(band, 0, False, sign)])
        self._apply_plaq_right(R, 1, self.plaq_mats[(band, 1, True, sign)])
        return R

    # ------------------------------------------------------------------ e^{sign dtau V} per site
    def evMatrix(self, sign, phivec, coshT, sinhT, coshC=1.0, sinhC=0.0):
        """detsdwopdim.cpp:3188-3229 (coshC = 1, sinhC = 0 while cdwU == 0)."""
        M = self.MSF
        ev = np.zeros((M, M), dtype=complex)
        p0 = phivec[0]
        p1 = phivec[1] if self.OPDIM > 1 else 0.0
        ev[0, 0] = coshT * coshC - sign * sinhC
        ev[1, 1] = coshT * coshC + sign * sinhC
        sinhT = sinhT * coshC
        ev[0, 1] = sign * (p0 - 1j * p1) * sinhT
        ev[1, 0] = sign * (p0 + 1j * p1) * sinhT
        if self.OPDIM == 3:
            p2 = phivec[2]
            ev[2, 2] = ev[0, 0]
            ev[3, 3] = ev[1, 1]
            ev[0, 3] = sign * p2 * sinhT
            ev[3, 0] = sign * p2 * sinhT
            ev[2, 1] = -sign * p2 * sinhT
            ev[1, 2] = -sign * p2 * sinhT
            ev[3, 2] = sign * (p0 - 1j * p1) * sinhT
            ev[2, 3] = sign * (p0 + 1j * p1) * sinhT
        return ev

    def _V_slice(self, sign, k):
        """All-site version: array [MSF, MSF, N] of e^{sign dtau V(phi_k)} entries
        (the vectors cd, cmd, mbx, mbcx, ax, max of detsdwopdim.cpp:2001-2030 / :2100-2130)."""
        M, N = self.MSF, self.N
        V = np.zeros((M, M, N), dtype=complex)
        c = self.coshTermPhi[k]
        x = self.sinhTermPhi[k]
        cd, cmd = c, c
        if self.pars.cdwU:           # cd / cmd and the cosh factor of the off-diagonal vectors, :2003-2030
            cd = c * self.coshTermCDWl[k] - sign * self.sinhTermCDWl[k]
            cmd = c * self.coshTermCDWl[k] + sign * self.sinhTermCDWl[k]
            x = x * self.coshTermCDWl[k]
        p0 = self.phi[k, :, 0]
        p1 = self.phi[k, :, 1] if self.OPDIM > 1 else np.zeros(N)
        b = (p0 - 1j * p1) * x
        bc = (p0 + 1j * p1) * x
        V[0, 0] = cd
        V[1, 1] = cmd
        V[0, 1] = sign * b
        V[1, 0] = sign * bc
        if self.OPDIM == 3:
            ax = self.phi[k, :, 2] * x
            V[2, 2] = cd
            V[3, 3] = cmd
            V[0, 3] = sign * ax
            V[3, 0] = sign * ax
            V[1, 2] = -sign * ax
            V[2, 1] = -sign * ax
            V[3, 2] = sign * b
            V[2, 3] = sign * bc
        return V

    def _blk(self, b):
        return slice(b * self.N, (b + 1) * self.N)

    # ------------------------------------------------------------------ B-multiplies (a10)
    def leftMultiplyBk(self, A, k):
        """detsdwopdim.cpp:1996-2070: B_k A, B_k = e^{-dtau V_k} diag(e^{dtau mu_band}) e^{-dtau K}."""
        M = self.MSF
        V = self._V_slice(-1, k)
        T = [math.exp(self.dtau * self.mu_band[c % 2]) * self.cbLMultHoppingExp(A[self._blk(c)], c % 2, -1)
             for c in range(M)]
        R = np.zeros_like(A, dtype=complex)
        for r in range(M):
            for c in range(M):
                R[self._blk(r)] += V[r, c][:, None] * T[c]
        return R

    def leftMultiplyBkInv(self, A, k):
        """detsdwopdim.cpp:2095-2167: B_k^{-1} A."""
        M = self.MSF
        V = self._V_slice(+1, k)
        R = np.zeros_like(A, dtype=complex)
        for r in range(M):
            S = np.zeros((self.N, A.shape[1]), dtype=complex)
            for c in range(M):
                S += V[r, c][:, None] * A[self._blk(c)]
            S *= math.exp(-self.dtau * self.mu_band[r % 2])
            R[self._blk(r)] = self.cbLMultHoppingExp(S, r % 2, +1)
        return R

    def rightMultiplyBk(self, A, k):
        """detsdwopdim.cpp:2190-2303: A B_k."""
        M = self.MSF
        V = self._V_slice(-1, k)
        R = np.zeros_like(A, dtype=complex)
        for c in range(M):
            S = np.zeros((A.shape[0], self.N), dtype=complex)
            for r in range(M):
                S += A[:, self._blk(r)] * V[r, c][None, :]
            S *= math.exp(self.dtau * self.mu_band[c % 2])
            R[:, self._blk(c)] = self.cbRMultHoppingExp(S, c % 2, -1)
        return R

    def rightMultiplyBkInv(self, A, k):
        """detsdwopdim.cpp:2328-2402: A B_k^{-1}."""
        M = self.MSF
        V = self._V_slice(+1, k)
        T = [math.exp(-self.dtau * self.mu_band[r % 2]) * self.cbRMultHoppingExp(A[:, self._blk(r)], r % 2, +1)
             for r in range(M)]
        R = np.zeros_like(A, dtype=complex)
        for c in range(M):
            for r in range(M):
                R[:, self._blk(c)] += T[r] * V[r, c][None, :]
        return R

    # chains (a14), detsdwopdim.cpp:2076-2090, 2172-2186, 2307-2324, 2406-2420
    def leftMultiplyBmat(self, A, k2, k1):
        if not self.pars.checkerboard:          # CB_NONE functors, detsdwopdim.h:1305-1375
            return self.computeBmatSDW(k2, k1) @ A
        R = A
        for k in range(k1 + 1, k2 + 1):
            R = self.leftMultiplyBk(R, k)
        return R

    def leftMultiplyBmatInv(self, A, k2, k1):
        if not self.pars.checkerboard:
            return np.linalg.inv(self.computeBmatSDW(k2, k1)) @ A
        R = A
        for k in range(k2, k1, -1):
            R = self.leftMultiplyBkInv(R, k)
        return R

    def rightMultiplyBmat(self, A, k2, k1):
        if not self.pars.checkerboard:
            return A @ self.computeBmatSDW(k2, k1)
        R = A
        for k in range(k2, k1, -1):
            R = self.rightMultiplyBk(R, k)
        return R

    def rightMultiplyBmatInv(self, A, k2, k1):
        if not self.pars.checkerboard:
            return A @ np.linalg.inv(self.computeBmatSDW(k2, k1))
        R = A
        for k in range(k1 + 1, k2 + 1):
            R = self.rightMultiplyBkInv(R, k)
        return R

    # ------------------------------------------------------------------ Green's function (a3, a4)
    def greenFromUdV(self, UdV_l, UdV_r):
        """detmodel.h:769-818."""
        VU_rl = UdV_r.V_t.conj().T @ UdV_l.U
        UtVt_rl = UdV_r.U.conj().T @ UdV_l.V_t
        tmp = udvDecompose(UtVt_rl + (UdV_r.d[:, None] * VU_rl) * UdV_l.d[None, :])
        Vt_product = UdV_l.V_t @ tmp.V_t
        U_product = UdV_r.U @ tmp.U
        g = (Vt_product * (1.0 / tmp.d)[None, :]) @ U_product.conj().T
        return g, tmp.d

    def greenFromEye_and_UdV(self, UdV_r):
        """detmodel.h:823-860."""
        tmp = udvDecompose(UdV_r.U.conj().T @ UdV_r.V_t + np.diag(UdV_r.d))
        Vt_product = UdV_r.V_t @ tmp.V_t
        U_product = UdV_r.U @ tmp.U
        g = (Vt_product * (1.0 / tmp.d)[None, :]) @ U_product.conj().T
        return g, tmp.d

    def setupUdVStorage_and_calculateGreen(self):
        """detmodel.h:680-713."""
        n, s, m = self.n, self.s, self.m
        eye = np.eye(self.ng, dtype=complex)
        storage = [None] * (n + 1)
        storage[0] = UdV.eye(self.ng)
        storage[1] = udvDecompose(self.leftMultiplyBmat(eye, s, 0))
        for l in range(1, n):
            k_l = s * l
            k_lp1 = s * (l + 1) if l < n - 1 else m
            BU = self.leftMultiplyBmat(storage[l].U, k_lp1, k_l)
            nxt = udvDecompose(BU * storage[l].d[None, :])
            nxt.V_t = storage[l].V_t @ nxt.V_t
            storage[l + 1] = nxt
        self.UdVStorage = storage
        self.g, self.g_inv_sv = self.greenFromEye_and_UdV(storage[n])
        self.currentTimeslice = m
        self.lastSweepDir = +1

    # ------------------------------------------------------------------ advance / wrap (a5-a7)
    def advanceDownGreen(self, l):
        """detmodel.h:956-1017."""
        storage, n, s, m = self.UdVStorage, self.n, self.s, self.m
        assert self.currentTimeslice == s * (l - 1)
        k_l = s * l if l < n else m
        k_lm1 = s * (l - 1)
        if l < n:
            st = storage[l]
            UdV_L = udvDecompose(st.d[:, None] * self.rightMultiplyBmat(st.V_t.conj().T, k_l, k_lm1))
            UdV_L.U = st.U @ UdV_L.U
        else:
            UdV_L = udvDecompose(self.rightMultiplyBmat(np.eye(self.ng, dtype=complex), k_l, k_lm1))
        if l - 1 > 0:
            self.g, self.g_inv_sv = self.greenFromUdV(UdV_L, storage[l - 1])
        else:
            self.g, self.g_inv_sv = self.greenFromEye_and_UdV(UdV_L)
        storage[l - 1] = UdV_L
        self.currentTimeslice = s * (l - 1)

    def advanceUpGreen(self, l):
        """detmodel.h:1109-1163."""
        storage, n, s, m = self.UdVStorage, self.n, self.s, self.m
        k_l = s * l
        k_lp1 = s * (l + 1) if l < n - 1 else m
        assert self.currentTimeslice == k_lp1
        st = storage[l]
        tmp = udvDecompose(self.leftMultiplyBmat(st.U, k_lp1, k_l) * st.d[None, :])
        tmp.V_t = st.V_t @ tmp.V_t
        if k_lp1 != m:
            self.g, self.g_inv_sv = self.greenFromUdV(storage[l + 1], tmp)
        else:
            self.g, self.g_inv_sv = self.greenFromEye_and_UdV(tmp)
        storage[l + 1] = tmp
        self.currentTimeslice = k_lp1

    def wrapDownGreen(self, k):
        """detmodel.h:1066-1095: G <- B_k^{-1} (G B_k)."""
        assert self.currentTimeslice == k
        self.g = self.leftMultiplyBmatInv(self.rightMultiplyBmat(self.g, k, k - 1), k, k - 1)
        self.currentTimeslice = k - 1

    def wrapUpGreen(self, k):
        """detmodel.h:1236-1259: G <- B_{k+1} (G B_{k+1}^{-1})."""
        assert self.currentTimeslice == k
        self.g = self.leftMultiplyBmat(self.rightMultiplyBmatInv(self.g, k + 1, k), k + 1, k)
        self.currentTimeslice = k + 1

    # ------------------------------------------------------------------ local updates (a17-a20)
    def deltaSPhi(self, site, k, newphi):
        """detsdwopdim.cpp:4186-4239."""
        p = self.pars
        dtau, r, u, c = self.dtau, p.r, p.u, p.c
        z = 4
        oldphi = self.phi[k, site]
        phiDiff = newphi - oldphi
        oldphiSq = float(np.dot(oldphi, oldphi))
        newphiSq = float(np.dot(newphi, newphi))
        phiSqDiff = newphiSq - oldphiSq
        if p.phi2bosons:
            return dtau * 0.5 * r * phiSqDiff
        phiPow4Diff = newphiSq * newphiSq - oldphiSq * oldphiSq
        m = self.m
        kEarlier = k - 1 if k > 1 else m       # PeriodicChainNearestNeighbors over slices 1..m
        kLater = k + 1 if k < m else 1
        phiTimeNeigh = self.phi[kLater, site] + self.phi[kEarlier, site]
        phiSpaceNeigh = np.zeros(self.OPDIM)
        for d in range(4):
            phiSpaceNeigh = phiSpaceNeigh + self.phi[k, self.neigh[d, site]]
        delta1 = (1.0 / (c * c * dtau)) * (phiSqDiff - float(np.dot(phiTimeNeigh, phiDiff)))
        delta2 = 0.5 * dtau * (z * phiSqDiff - 2.0 * float(np.dot(phiSpaceNeigh, phiDiff)))
        delta3 = dtau * (0.5 * r * phiSqDiff + 0.25 * u * phiPow4Diff)
        return delta1 + delta2 + delta3

    def get_delta_forsite(self, newphi, k, site, new_cdwl=None):
        """detsdwopdim.cpp:3179-3289: e^{-dtau V_new} e^{+dtau V_old} - 1 at one site."""
        cCo, sCo, cCn, sCn = 1.0, 0.0, 1.0, 0.0
        if self.pars.cdwU:
            cCo, sCo = self.coshTermCDWl[k, site], self.sinhTermCDWl[k, site]
            cCn, sCn = self.getCoshSinhTermCDWl(self.cdwl[k, site] if new_cdwl is None else new_cdwl)
        evOld = self.evMatrix(+1, self.phi[k, site], self.coshTermPhi[k, site], self.sinhTermPhi[k, site], cCo, sCo)
        cN, sN = self.getCoshSinhTermPhi(newphi)
        emvNew = self.evMatrix(-1, newphi, cN, sN, cCn, sCn)
        return emvNew @ evOld - np.eye(self.MSF)

    def proposeNewPhiBox(self, site, k):
        """detsdwopdim.cpp:3922-3931."""
        newphi = self.phi[k, site].copy()
        for d in range(self.OPDIM):
            newphi[d] += self.rng.randRange(-self.phiDelta, +self.phiDelta)
        return newphi

    def _rotated(self, vec, r_new_over_r=None, new_r=None):
        """the rotation shared by proposeRandomRotatedVector<3> and proposeRandomRotatedScaledVector<3> (detsdwopdim.cpp:3945-3992,
        4112-4146): new direction in a cone around the old one, cos(theta) in [angleDelta, 1]; length r (rotate) or new_r"""
        x, y, z = float(vec[0]), float(vec[1]), float(vec[2])
        # pow(x, 2.0): g++ -O2 expands it to x * x (exact, allowed without -ffast-math), while libm's pow is only guaranteed to ~0.52 ulp
        # -- the two differ in the last bit once in ~1e4 arguments, which the 12-sweep fixture is long enough to see
        x2, y2, z2 = x * x, y * y, z * z
        r2 = x2 + y2 + z2
        r = math.sqrt(r2)
        cosTheta = self.rng.rand01() * (1.0 - self.angleDelta) + self.angleDelta
        phi = self.rng.rand01() * 2.0 * math.pi
        sinTheta = math.sqrt(1.0 - cosTheta * cosTheta)
        sinPhi, cosPhi = _sincos(phi)
        x2n, y2n = x2 / r2, y2 / r2
        xn, yn, zn = x / r, y / r, z / r
        newx = (sinTheta / (x2n + y2n)) * ((x2n * zn + y2n) * cosPhi + (zn - 1) * xn * yn * sinPhi) + xn * cosTheta
        newy = (sinTheta / (x2n + y2n)) * ((zn - 1) * xn * yn * cosPhi + (x2n + y2n * zn) * sinPhi) + yn * cosTheta
        newz = -sinTheta * (xn * cosPhi + yn * sinPhi) + zn * cosTheta
        length = r if new_r is None else new_r
        return np.array([newx * length, newy * length, newz * length])

    def proposeRotatedPhi(self, site, k):
        """detsdwopdim.cpp:3945-4002: two uniforms; always a valid proposal"""
        return self._rotated(self.phi[k, site]), True

    def proposeScaledPhi(self, site, k):
        """detsdwopdim.cpp:4016-4077: new |phi|^3 Gaussian around the old one (width scaleDelta); not positive -> changed = NONE"""
        x, y, z = (float(v) for v in self.phi[k, site])
        x2, y2, z2 = x * x, y * y, z * z
        r3 = math.pow(x2 + y2 + z2, 3.0 / 2.0)
        new_r3 = self.normal_distribution.get(self.scaleDelta, r3)
        if new_r3 <= 0:
            return self.phi[k, site].copy(), False
        scale = math.pow(new_r3 / r3, 1.0 / 3.0)
        return np.array([x * scale, y * scale, z * scale]), True

    def proposeRotatedScaledPhi(self, site, k):
        """detsdwopdim.cpp:4092-4160: Gaussian draw for |phi|^3 first; only if it is positive the two uniforms of the rotation"""
        x, y, z = (float(v) for v in self.phi[k, site])
        x2, y2, z2 = x * x, y * y, z * z
        r = math.sqrt(x2 + y2 + z2)
        r3 = math.pow(r, 3)
        new_r3 = self.normal_distribution.get(self.scaleDelta, r3)
        if new_r3 <= 0:
            return self.phi[k, site].copy(), False
        # (the rotation draws its uniforms before new_r = new_r3^(1/3) is formed, as in the reference)
        vec = self.phi[k, site]
        cos_draws = self._rotated(vec, new_r=1.0)               # unit vector
        new_r = math.pow(new_r3, 1.0 / 3.0)
        return cos_draws * new_r, True

    def proposeNewCDWl(self, site, k):
        """detsdwopdim.cpp:4173-4182: one uniform, the field phi stays."""
        return self.cdwl_from_uniform(self.rng.rand01())

    def updateInSlice_delayed(self, k, what="phi"):
        """detsdwopdim.cpp:3023-3175; what = "phi" (box proposals) or "cdwl" (proposeNewCDWl: changed == CDWL, probSPhi = 1)."""
        MSF, N, D = self.MSF, self.N, self.pars.delaySteps
        g = self.g
        accratio = 0.0
        eyeS = np.eye(MSF)
        site = 0
        while site < N:
            delayStepsNow = min(D, N - site)
            X = np.zeros((MSF * N, MSF * delayStepsNow), dtype=complex)
            Y = np.zeros((MSF * delayStepsNow, MSF * N), dtype=complex)
            j = 0
            while j < delayStepsNow and site < N:
                if what == "phi":
                    newphi = self.proposeNewPhiBox(site, k)
                    new_cdwl = int(self.cdwl[k, site])
                    probSPhi = math.exp(-self.deltaSPhi(site, k, newphi))
                elif what in ("rotate", "scale", "rotate_and_scale"):
                    newphi, valid = (self.proposeRotatedPhi if what == "rotate" else self.proposeScaledPhi if what == "scale"
                                     else self.proposeRotatedScaledPhi)(site, k)
                    if not valid:                       # changed == NONE (:3063): rejected at once, no acceptance draw
                        site += 1
                        continue
                    new_cdwl = int(self.cdwl[k, site])
                    probSPhi = math.exp(-self.deltaSPhi(site, k, newphi))
                else:
                    newphi = self.phi[k, site].copy()
                    new_cdwl = self.proposeNewCDWl(site, k)
                    probSPhi = 1.0
                delta = self.get_delta_forsite(newphi, k, site, new_cdwl)
                idx = site + N * np.arange(MSF)
                Rj = g[idx, :].copy()
                if j > 0:
                    Rj += X[idx, :MSF * j] @ Y[:MSF * j, :]
                Sj = Rj[:, idx]
                Mj = eyeS - Sj @ delta + delta
                det = np.linalg.det(Mj)
                if self.OPDIM == 3:
                    probSFermion = det.real
                else:
                    probSFermion = abs(det) ** 2
                prob_cdwl = self.cdwl_gamma(new_cdwl) / self.cdwl_gamma(int(self.cdwl[k, site]))     # :3110 (1 for a phi proposal)
                prob = probSPhi * probSFermion * prob_cdwl
                if what == "cdwl" and new_cdwl == int(self.cdwl[k, site]):
                    # A proposal that draws the value the site already has (one in four): delta = 0 and prob = 1 in exact arithmetic, so
                    # `prob > 1.0` is false, a uniform is drawn and the (null) update accepted.  In floating point the reference's
                    # |det|^2 is 1 or 1 + 2^-52 depending on the last bit of e^{-dtau V} e^{+dtau V} - 1 and of G -- about one null
                    # proposal in a hundred takes the other branch and skips the uniform; no independent implementation (or another BLAS
                    # under the reference itself) can follow that.  The oracle and the HIP path take the exact-arithmetic branch; the
                    # cdw fixtures use seeds on which the reference does too over their whole trajectory (oracle/make_golden.py).
                    prob = 1.0
                if prob > 1.0 or self.rng.rand01() < prob:
                    accratio += 1.0
                    self.phi[k, site] = newphi
                    self.cdwl[k, site] = new_cdwl
                    self.coshTermPhi[k, site], self.sinhTermPhi[k, site] = self.getCoshSinhTermPhi(newphi)
                    if self.pars.cdwU:
                        self.coshTermCDWl[k, site], self.sinhTermCDWl[k, site] = self.getCoshSinhTermCDWl(new_cdwl)
                    Cj = g[:, idx].copy()
                    if j > 0:
                        Cj += X[:, :MSF * j] @ Y[:MSF * j, idx]
                    Rj[np.arange(MSF), idx] -= 1.0
                    X[:, MSF * j:MSF * (j + 1)] = Cj @ delta
                    Y[MSF * j:MSF * (j + 1), :] = np.linalg.solve(Mj, Rj)
                    j += 1
                site += 1
            if j > 0:
                g += X[:, :MSF * j] @ Y[:MSF * j, :]
        return accratio / N

    def updateInSlice(self, k):
        """detsdwopdim.cpp:2428-2489 (delayed method)."""
        self.normal_distribution.reset()
        for _ in range(self.pars.repeatUpdateInSlice):
            spm = self.pars.spinProposalMethod
            if spm == "box":
                what = "phi"
            elif spm == "rotate_then_scale":            # each sweep alternates between rotating and scaling (:2447-2462)
                what = "rotate" if self.performedSweeps % 2 == 0 else "scale"
            else:
                what = "rotate_and_scale"
            self.lastAccRatioLocal_phi = self.updateInSlice_delayed(k, what)
        if self.pars.cdwU:                              # :2474-2485: second pass over the slice, its acceptance ratio is discarded
            self.updateInSlice_delayed(k, "cdwl")

    def updateInSliceThermalization(self, k):
        """detsdwopdim.cpp:3294-3375."""
        self.updateInSlice(k)
        spm = self.pars.spinProposalMethod
        if spm == "box":
            what, ra = "box", self.accRatioLocal_box_RA
        elif spm == "rotate_then_scale":                # must match the order of moves in updateInSlice (:3303-3311)
            what = "rotate" if self.performedSweeps % 2 == 0 else "scale"
        else:                                           # alternate every AccRatioAdjustmentSamples sweeps (:3312-3320)
            what = "rotate" if self.performedSweeps % (2 * self.AccRatioAdjustmentSamples) < self.AccRatioAdjustmentSamples else "scale"
        if what == "rotate":
            ra = self.accRatioLocal_rotate_RA
        elif what == "scale":
            ra = self.accRatioLocal_scale_RA
        ra.addValue(self.lastAccRatioLocal_phi)
        if ra.samplesAdded % self.AccRatioAdjustmentSamples == 0:
            avg = ra.get()
            tgt = self.targetAccRatioLocal_phi
            if what == "box":
                if avg < tgt:
                    self.phiDelta *= self.phiDeltaShrinkFactor
                elif avg > tgt:
                    self.phiDelta *= self.phiDeltaGrowFactor
            elif what == "rotate":                      # :3344-3353 -- angleDelta = minimal cos(theta); bisection between cur min / max
                if avg < tgt and self.angleDelta < self.MaxAngleDelta:
                    self.curminAngleDelta = self.angleDelta
                    self.angleDelta += (self.curmaxAngleDelta - self.angleDelta) / 2
                elif avg > tgt and self.angleDelta > self.MinAngleDelta:
                    self.curmaxAngleDelta = self.angleDelta
                    self.angleDelta -= (self.angleDelta - self.curminAngleDelta) / 2
            elif self.pars.adaptScaleVariance:          # :3356-3372 (both branches test avg > target, as in the reference)
                if avg > tgt and self.scaleDelta < self.MaxScaleDelta:
                    self.curminScaleDelta = self.scaleDelta
                    self.scaleDelta += (self.curmaxScaleDelta - self.scaleDelta) / 2
                elif avg > tgt and self.scaleDelta > self.MinScaleDelta:
                    self.curmaxScaleDelta = self.scaleDelta
                    self.scaleDelta -= (self.scaleDelta - self.curminScaleDelta) / 2

    # ------------------------------------------------------------------ sweeps (a9)
    def sweepDown(self, upd):
        """detmodel.h:1333-1399."""
        n, s, m = self.n, self.s, self.m
        for k in range(m, (n - 1) * s, -1):
            assert self.currentTimeslice == k
            upd(k)
            self.wrapDownGreen(k)
        for l in range(n - 1, 0, -1):
            self.advanceDownGreen(l + 1)
            for k in range(l * s, (l - 1) * s, -1):
                upd(k)
                self.wrapDownGreen(k)
        self.advanceDownGreen(1)

    def sweepUp(self, upd):
        """detmodel.h:1266-1325."""
        n, s, m = self.n, self.s, self.m
        self.UdVStorage[0] = UdV.eye(self.ng)
        for l in range(0, n - 1):
            for k in range(l * s + 1, (l + 1) * s + 1):
                self.wrapUpGreen(k - 1)
                upd(k)
            self.advanceUpGreen(l)
        for k in range((n - 1) * s + 1, m + 1):
            self.wrapUpGreen(k - 1)
            upd(k)
        self.advanceUpGreen(n - 1)

    def _sweep(self, upd):
        """detmodel.h:1408-1478."""
        if self.lastSweepDir == +1:
            self.globalMove()
            self.sweepDown(upd)
            self.lastSweepDir = -1
        else:
            self.sweepUp(upd)
            self.lastSweepDir = +1
        self.performedSweeps += 1

    def sweepThermalization(self):
        """detsdwopdim.cpp:4474-4502."""
        self._sweep(self.updateInSliceThermalization)

    def sweep(self, takeMeasurements=False):
        """detsdwopdim.cpp:4423-4471; with takeMeasurements the bosonic observables of initMeasurements / measure /
        finishMeasurements (:441-456, :509-545, :903-921) -- the reference run with turnoffFermionMeasurements; the
        fermionic observables are SURVEY 8f."""
        if not takeMeasurements:
            self._sweep(self.updateInSlice)
            return
        self.initMeasurements()

        def update_and_measure(k):               # updateInSliceAndMaybeMeasure, detmodel.h:1279-1285, 1346-1352
            self.updateInSlice(k)
            self.measure(k)
        self._sweep(update_and_measure)
        self.finishMeasurements()

    @staticmethod
    def _adot(a, b):
        v1 = v2 = 0.0
        n = len(a)
        i = 0
        while i + 1 < n:
            v1 += float(a[i]) * float(b[i])
            v2 += float(a[i + 1]) * float(b[i + 1])
            i += 2
        if i < n:
            v1 += float(a[i]) * float(b[i])
        return v1 + v2

    # ------------------------------------------------------------------ fermionic observables
    def shiftGreenSymmetric(self):
        """detsdwopdim.cpp:4507-4612: e^{-dtau K/2} G e^{+dtau K/2} with the checkerboard half steps
        (right: sub 1 then sub 0 with +sinh; left: sub 1 then sub 0 with -sinh), or the dense half propagators."""
        N, M = self.N, self.MSF
        g = self.g
        tmp = np.zeros_like(g)
        new = np.zeros_like(g)
        if not self.pars.checkerboard:
            self._dense_propK()
        for row in range(M):
            for col in range(M):
                band = col % 2
                blk = np.array(g[self._blk(row), self._blk(col)], dtype=complex, copy=True)
                if self.pars.checkerboard:
                    self._apply_plaq_right(blk, 1, self.plaq_mats[(band, 1, True, +1)])
                    self._apply_plaq_right(blk, 0, self.plaq_mats[(band, 0, True, +1)])
                else:
                    blk = blk @ self._propK_half_inv[band]
                tmp[self._blk(row), self._blk(col)] = blk
        for col in range(M):
            for row in range(M):
                band = row % 2
                blk = np.array(tmp[self._blk(row), self._blk(col)], dtype=complex, copy=True)
                if self.pars.checkerboard:
                    self._apply_plaq_left(blk, 1, self.plaq_mats[(band, 1, True, -1)])
                    self._apply_plaq_left(blk, 0, self.plaq_mats[(band, 0, True, -1)])
                else:
                    blk = self._propK_half[band] @ blk
                new[self._blk(row), self._blk(col)] = blk
        return new

    def _gl1_blocks(self, gs):
        """gl1 (detsdwopdim.cpp:594-612) as a 4 x 4 table of N x N blocks indexed by BandSpin XUP=0, YDOWN=1, XDOWN=2,
        YUP=3; for OPDIM < 3 the lower 2 x 2 sector is the complex conjugate of the upper one, the rest is zero."""
        N = self.N
        B = [[None] * 4 for _ in range(4)]
        for b1 in range(4):
            for b2 in range(4):
                if self.OPDIM == 3:
                    B[b1][b2] = gs[b1 * N:(b1 + 1) * N, b2 * N:(b2 + 1) * N]
                elif b1 < 2 and b2 < 2:
                    B[b1][b2] = gs[b1 * N:(b1 + 1) * N, b2 * N:(b2 + 1) * N]
                elif b1 >= 2 and b2 >= 2:
                    B[b1][b2] = np.conj(gs[(b1 - 2) * N:(b1 - 1) * N, (b2 - 2) * N:(b2 - 1) * N])
                else:
                    B[b1][b2] = np.zeros((N, N), dtype=complex)
        return B

    def measureFermionic(self, k):
        """measure(), fermionic part (detsdwopdim.cpp:545-899)."""
        N, L = self.N, self.L
        gs = self.shiftGreenSymmetric()
        if self.OPDIM == 3:
            self.greenK0 += float(np.real(np.sum(gs)))
            self.greenLocal += float(np.real(np.trace(gs))) / (4.0 * N)
        else:
            self.greenK0 += 2.0 * float(np.real(np.sum(gs)))
            self.greenLocal += 2.0 * float(np.real(np.trace(gs))) / (4.0 * N)
        B = self._gl1_blocks(gs)
        XUP, YDOWN, XDOWN, YUP = 0, 1, 2, 3

        def bs(band, spin):          # getBandSpin (detsdwopdim.h:268-273); spin: 0 up, 1 down
            if band == XBAND:
                return XUP if spin == 0 else XDOWN
            return YUP if spin == 0 else YDOWN
        UP, DN = 0, 1
        # k-space occupation (:616-659)
        p = self.pars
        offx = 0.5 if p.bc in ("apbc-x", "apbc-xy") else 0.0
        offy = 0.5 if p.bc in ("apbc-y", "apbc-xy") else 0.0
        ix = np.arange(N) % L
        iy = np.arange(N) // L
        dx = (ix[:, None] - ix[None, :]).astype(float)
        dy = (iy[:, None] - iy[None, :]).astype(float)
        gx = B[XUP][XUP] + B[XDOWN][XDOWN]
        gy = B[YUP][YUP] + B[YDOWN][YDOWN]
        for ksite in range(N):
            ky = -math.pi + (float(ksite // L) + offy) * 2 * math.pi / float(L)
            kx = -math.pi + (float(ksite % L) + offx) * 2 * math.pi / float(L)
            phase = np.exp(1j * (kx * dx + ky * dy))
            self.kOccX[ksite] += float(np.real(np.sum(phase * gx)))
            self.kOccY[ksite] += float(np.real(np.sum(phase * gy)))
        # equal-time pairing correlations (:661-722)
        for i in range(N):
            plus = 0j
            minus = 0j
            for (A, Bs) in ((i, 0), (0, i)):
                def gl(b1, s1, b2, s2):
                    return complex(B[bs(b1, s1)][bs(b2, s2)][A, Bs])
                X, Y = XBAND, YBAND
                t = [gl(X, DN, X, UP) * gl(X, UP, X, DN), gl(X, DN, X, DN) * gl(X, UP, X, UP),
                     gl(X, DN, Y, UP) * gl(X, UP, Y, DN), gl(X, DN, Y, DN) * gl(X, UP, Y, UP),
                     gl(Y, DN, X, UP) * gl(Y, UP, X, DN), gl(Y, DN, X, DN) * gl(Y, UP, X, UP),
                     gl(Y, DN, Y, UP) * gl(Y, UP, Y, DN), gl(Y, DN, Y, DN) * gl(Y, UP, Y, UP)]
                plus += -4.0 * (t[0] - t[1] + t[2] - t[3] + t[4] - t[5] + t[6] - t[7])
                minus += -4.0 * (t[0] - t[1] - t[2] + t[3] - t[4] + t[5] + t[6] - t[7])
            self.pairPlus[i] += plus.real
            self.pairMinus[i] += minus.real
        # occDiffSq (:866-897)
        contrib = 0j
        X, Y = XBAND, YBAND
        for i in range(N):
            def gl(b1, s1, b2, s2):
                return complex(B[bs(b1, s1)][bs(b2, s2)][i, i])
            contrib += (-2.0 * gl(X, DN, X, UP) * gl(X, UP, X, DN) + gl(X, UP, X, UP)
                        + 2.0 * gl(X, DN, Y, DN) * gl(Y, DN, X, DN)
                        + 2.0 * gl(X, UP, Y, DN) * gl(Y, DN, X, UP)
                        + gl(Y, DN, Y, DN)
                        - 2.0 * gl(X, UP, X, UP) * gl(Y, DN, Y, DN)
                        + 2.0 * gl(X, DN, Y, UP) * gl(Y, UP, X, DN)
                        + 2.0 * gl(X, UP, Y, UP) * gl(Y, UP, X, UP)
                        - 2.0 * gl(Y, DN, Y, UP) * gl(Y, UP, Y, DN)
                        + gl(X, DN, X, DN) * (1.0 + 2.0 * gl(X, UP, X, UP) - 2.0 * gl(Y, DN, Y, DN) - 2.0 * gl(Y, UP, Y, UP))
                        + gl(Y, UP, Y, UP)
                        - 2.0 * gl(X, UP, X, UP) * gl(Y, UP, Y, UP)
                        + 2.0 * gl(Y, DN, Y, DN) * gl(Y, UP, Y, UP))
        self.occDiffSq += contrib.real / float(N)

    def initMeasurements(self):
        N = self.N
        self.greenK0 = 0.0
        self.greenLocal = 0.0
        self.kOccX = np.zeros(N)
        self.kOccY = np.zeros(N)
        self.pairPlus = np.zeros(N)
        self.pairMinus = np.zeros(N)
        self.pairPlusMax = 0.0
        self.pairMinusMax = 0.0
        self.occDiffSq = 0.0
        self.meanPhi = np.zeros(self.OPDIM)
        self.normMeanPhi = 0.0
        self.phiRhoS_Gs = 0.0
        self.phiRhoS_Gc = 0.0
        self.associatedEnergy = 0.0
        self._measured_slices = set()

    def measure(self, k):
        self._measured_slices.add(k)
        phi = self.phi
        if self.OPDIM == 2:
            for site in range(self.N):
                ps = phi[k, site]
                px = phi[k, int(self.neigh[0, site])]      # XPLUS
                py = phi[k, int(self.neigh[2, site])]      # YPLUS
                self.phiRhoS_Gc += self._adot(ps, px) + self._adot(ps, py)
                self.phiRhoS_Gs += float(px[0]) * float(ps[1]) - float(px[1]) * float(ps[0])
        for site in range(self.N):
            ps = phi[k, site]
            self.meanPhi = self.meanPhi + ps
            self.associatedEnergy += self._adot(ps, ps)
        if not self.pars.turnoffFermionMeasurements:
            self.measureFermionic(k)

    def finishMeasurements(self):
        N, m = self.N, self.m
        assert len(self._measured_slices) == m
        self.meanPhi = self.meanPhi / float(N * m)
        self.normMeanPhi = math.sqrt(sum(float(x) * float(x) for x in self.meanPhi))
        if self.OPDIM == 2:
            self.phiRhoS_Gc *= (0.5 * self.dtau)
            self.phiRhoS_Gs *= self.dtau
        self.associatedEnergy /= (2.0 * N * m)
        if not self.pars.turnoffFermionMeasurements:            # detsdwopdim.cpp:923-1015
            L = self.L
            self.greenK0 /= float(m)
            self.greenLocal /= float(m)
            self.kOccX = 2.0 - self.kOccX / float(m * N)
            self.kOccY = 2.0 - self.kOccY / float(m * N)
            self.pairPlus = self.pairPlus / m
            self.pairMinus = self.pairMinus / m
            far = [(L // 2 + ox) + (L // 2 + oy) * L for oy in (-1, 0, 1) for ox in (-1, 0, 1)]   # coordsToSite(x, y)
            self.pairPlusMax = sum(float(self.pairPlus[i]) for i in far) / 9.0
            self.pairMinusMax = sum(float(self.pairMinus[i]) for i in far) / 9.0
            self.occDiffSq /= float(m)

    # ------------------------------------------------------------------ global shift move (a21)
    def phiAction(self):
        """detsdwopdim.cpp:4242-4300."""
        p = self.pars
        dtau, r, u, c, m = self.dtau, p.r, p.u, p.c, self.m
        phi = self.phi
        action = 0.0
        for k in range(1, m + 1):
            kprev = k - 1 if k > 1 else m
            for site in range(self.N):
                ph = phi[k, site]
                if not p.phi2bosons:
                    td = (ph - phi[kprev, site]) / dtau
                    action += (dtau / (2.0 * c * c)) * float(np.dot(td, td))
                    xd = ph - phi[k, self.neigh[0, site]]
                    action += 0.5 * dtau * float(np.dot(xd, xd))
                    yd = ph - phi[k, self.neigh[2, site]]
                    action += 0.5 * dtau * float(np.dot(yd, yd))
                phisq = float(np.dot(ph, ph))
                action += 0.5 * dtau * r * phisq
                if not p.phi2bosons:
                    action += 0.25 * dtau * u * phisq ** 2
        return action

    def globalMove(self):
        """detsdwopdim.cpp:3461-3486."""
        p = self.pars
        if self.performedSweeps % p.globalUpdateInterval == 0:
            if p.globalShift:
                self.attemptGlobalShiftMove()
            if p.wolffClusterUpdate:
                self.attemptWolffClusterUpdate()
            if p.wolffClusterShiftUpdate:
                self.attemptWolffClusterShiftUpdate()

    def _fermion_ratio(self, old_sv):
        log_prob = float(np.sum(np.log(self.g_inv_sv) - np.log(old_sv)))
        prob_fermion = math.exp(log_prob)
        if self.OPDIM < 3:
            prob_fermion = prob_fermion ** 2
        return prob_fermion

    def _backup(self):
        return (self.phi.copy(), self.coshTermPhi.copy(), self.sinhTermPhi.copy(), self.g, self.g_inv_sv,
                self.UdVStorage)

    def _restore(self, bak):
        (self.phi, self.coshTermPhi, self.sinhTermPhi, self.g, self.g_inv_sv, self.UdVStorage) = bak

    def buildAndFlipCluster(self):
        """detsdwopdim.cpp:3806-3883 with randomDirection<OPDIM> (:3765-3803).  The cosh/sinh terms are refreshed
        for the whole field by the callers here (the reference updates them site by site when asked to)."""
        p, m, N, dtau = self.pars, self.m, self.N, self.dtau
        if self.OPDIM == 1:
            rd = np.array([-1.0 if self.rng.rand01() <= 0.5 else +1.0])
        elif self.OPDIM == 2:
            rd = np.array(self.rng.randPointOnCircle())
        else:
            rd = np.array(self.rng.randPointOnSphere())
        phi = self.phi

        def adot(a, b):
            # arma::dot on short vectors (op_dot::direct_dot_arma): two accumulators over even / odd elements, no fma
            v1 = v2 = 0.0
            n = len(a)
            i = 0
            while i + 1 < n:
                v1 += float(a[i]) * float(b[i])
                v2 += float(a[i + 1]) * float(b[i + 1])
                i += 2
            if i < n:
                v1 += float(a[i]) * float(b[i])
            return v1 + v2

        def projected(site, k):
            return adot(phi[k, site], rd)

        def flip(site, k):
            ph = phi[k, site]
            f = 2.0 * adot(ph, rd)
            phi[k, site] = np.array([float(ph[d]) - f * float(rd[d]) for d in range(self.OPDIM)])

        visited = np.zeros((N, m + 1), dtype=bool)
        k = self.rng.randInt(1, m)
        site = self.rng.randInt(0, N - 1)
        flip(site, k)
        visited[site, k] = True
        stack = [(site, k)]
        cluster_size = 1
        while stack:
            site, k = stack.pop()
            for d in range(4):                                   # XPLUS, XMINUS, YPLUS, YMINUS
                nb = int(self.neigh[d, site])
                if not visited[nb, k]:
                    bond_arg = 2.0 * dtau * projected(site, k) * projected(nb, k)
                    if bond_arg < 0 and self.rng.rand01() <= (1.0 - math.exp(bond_arg)):
                        flip(nb, k)
                        visited[nb, k] = True
                        stack.append((nb, k))
                        cluster_size += 1
            for kn in ((k + 1 if k < m else 1), (k - 1 if k > 1 else m)):      # ChainDir PLUS, MINUS over 1..m
                if not visited[site, kn]:
                    bond_arg = (2.0 / dtau) * projected(site, k) * projected(site, kn)
                    if bond_arg < 0 and self.rng.rand01() <= (1.0 - math.exp(bond_arg)):
                        flip(site, kn)
                        visited[site, kn] = True
                        stack.append((site, kn))
                        cluster_size += 1
        return cluster_size

    def attemptWolffClusterUpdate(self):
        """detsdwopdim.cpp:3488-3562."""
        assert self.currentTimeslice == self.m
        bak = self._backup()
        old_sv = self.g_inv_sv
        sizes = [self.buildAndFlipCluster() for _ in range(self.pars.repeatWolffPerSweep)]
        self.updateCoshSinhTermsPhi()
        self.setupUdVStorage_and_calculateGreen()
        prob_fermion = self._fermion_ratio(old_sv)
        self.attemptedWolffClusterUpdates += 1
        if prob_fermion >= 1.0 or self.rng.rand01() < prob_fermion:
            self.acceptedWolffClusterUpdates += 1
            self.addedWolffClusterSize += float(sum(sizes))
        else:
            self._restore(bak)

    def attemptWolffClusterShiftUpdate(self):
        """detsdwopdim.cpp:3647-3751."""
        assert self.currentTimeslice == self.m
        bak = self._backup()
        old_sv = self.g_inv_sv
        sizes = [self.buildAndFlipCluster() for _ in range(self.pars.repeatWolffPerSweep)]
        old_action = self.phiAction()                # after the cluster flips
        for d in range(self.OPDIM):
            rr = self.rng.randRange(-self.phiDelta, +self.phiDelta)
            self.phi[:, :, d] += rr
        new_action = self.phiAction()
        prob_scalar = math.exp(-(new_action - old_action))
        self.updateCoshSinhTermsPhi()
        self.setupUdVStorage_and_calculateGreen()
        prob = prob_scalar * self._fermion_ratio(old_sv)
        self.attemptedWolffClusterShiftUpdates += 1
        if prob >= 1.0 or self.rng.rand01() < prob:
            self.acceptedWolffClusterShiftUpdates += 1
            self.addedWolffClusterSize += float(sum(sizes))
        else:
            self._restore(bak)

    def attemptGlobalShiftMove(self):
        """detsdwopdim.cpp:3565-3644, backups :3886-3917, displacement :3755-3763."""
        old_action = self.phiAction()
        assert self.currentTimeslice == self.m
        bak = (self.phi.copy(), self.coshTermPhi.copy(), self.sinhTermPhi.copy(), self.g, self.g_inv_sv,
               self.UdVStorage)
        old_sv = self.g_inv_sv
        for d in range(self.OPDIM):
            rr = self.rng.randRange(-self.phiDelta, +self.phiDelta)
            self.phi[:, :, d] += rr
        self.updateCoshSinhTermsPhi()
        self.setupUdVStorage_and_calculateGreen()
        new_action = self.phiAction()
        prob_scalar = math.exp(-(new_action - old_action))
        log_prob = float(np.sum(np.log(self.g_inv_sv) - np.log(old_sv)))
        prob_fermion = math.exp(log_prob)
        if self.OPDIM < 3:
            prob_fermion = prob_fermion ** 2
        prob = prob_scalar * prob_fermion
        self.attemptedGlobalShifts += 1
        if prob >= 1.0 or self.rng.rand01() < prob:
            self.acceptedGlobalShifts += 1
        else:
            (self.phi, self.coshTermPhi, self.sinhTermPhi, self.g, self.g_inv_sv, self.UdVStorage) = bak

    # ------------------------------------------------------------------ replica exchange (8e)
    def get_exchange_action_contribution(self):
        """detsdwopdim.cpp:5205-5216."""
        return 0.5 * self.dtau * float(np.sum(self.phi[1:] ** 2))

    # ------------------------------------------------------------------ dense B (a15, checks only)
    def _dense_propK(self):
        """setupPropK (detsdwopdim.cpp:1210-1285) + computePropagator (detmodel.cpp:31-39):
        propK[band] = exp(-dtau K_band) through eig_sym, K_band including -mu_band on the diagonal,
        the APBC signs and the Peierls phases of zmag[XUP] / zmag[YDOWN]."""
        if getattr(self, "_propK", None) is not None:
            return self._propK
        p, N, L = self.pars, self.N, self.L
        hop = {XBAND: (p.txhor, p.txver), YBAND: (p.tyhor, p.tyver)}
        props = {}
        for band in (XBAND, YBAND):
            K = -self.mu_band[band] * np.eye(N, dtype=complex)
            zm = (1.0 / N) if p.weakZflux else 0.0
            for site in range(N):
                sy, sx = divmod(site, L)
                for d in range(4):
                    nb = self.neigh[d, site]
                    h = hop[band][0] if d < 2 else hop[band][1]
                    if p.bc in ("apbc-x", "apbc-xy") and ((sx == 0 and d == 1) or (sx == L - 1 and d == 0)):
                        h *= -1
                    if p.bc in ("apbc-y", "apbc-xy") and ((sy == 0 and d == 3) or (sy == L - 1 and d == 2)):
                        h *= -1
                    ph = 1.0 + 0j
                    if d == 0:
                        ph = np.exp(1j * (-2.0 * math.pi * zm * sy))
                    if d == 1:
                        ph = np.exp(1j * (+2.0 * math.pi * zm * sy))
                    if d == 2 and sy == L - 1:
                        ph = np.exp(1j * (+2.0 * math.pi * zm * L * sx))
                    if d == 3 and sy == 0:
                        ph = np.exp(1j * (-2.0 * math.pi * zm * L * sx))
                    K[site, nb] -= h * ph
            ev, evec = np.linalg.eigh(K)
            props[band] = (evec * np.exp(-self.dtau * ev)) @ evec.conj().T
            # propK_half, propK_half_inv (setupPropK, detsdwopdim.cpp:1282-1283)
            self._propK_half = getattr(self, "_propK_half", {})
            self._propK_half_inv = getattr(self, "_propK_half_inv", {})
            self._propK_half[band] = (evec * np.exp(-0.5 * self.dtau * ev)) @ evec.conj().T
            self._propK_half_inv[band] = (evec * np.exp(+0.5 * self.dtau * ev)) @ evec.conj().T
        self._propK = props
        return props

    def computeBmatDense(self, k):
        """singleTimesliceProp of computeBmatSDW (detsdwopdim.cpp:1324-1474): e^{-dtau V_k} e^{-dtau K};
        block (r, c) = diag(V_k[r, c]) propK[band(c)].  As in the reference, the O(3) lower blocks use
        the same propK as the upper ones (its TODO at :1439 notes the flux case is not adapted; flux is
        only allowed for opdim = 2 anyway)."""
        props = self._dense_propK()
        V = self._V_slice(-1, k)
        B = np.zeros((self.ng, self.ng), dtype=complex)
        for r in range(self.MSF):
            for c in range(self.MSF):
                B[self._blk(r), self._blk(c)] = V[r, c][:, None] * props[c % 2]
        return B

    def computeBmatSDW(self, k2, k1):
        """detsdwopdim.cpp:1309-1485: B(k2, k1) = B_k2 B_{k2-1} ... B_{k1+1}, identity for k2 == k1."""
        if k2 == k1:
            return np.eye(self.ng, dtype=complex)
        assert k1 < k2 <= self.m
        R = self.computeBmatDense(k2)
        for k in range(k2 - 1, k1, -1):
            R = R @ self.computeBmatDense(k)
        return R
