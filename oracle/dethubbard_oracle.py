"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product package.

CPU restatement (numpy / scipy LAPACK) of the reference's Hubbard replica -- BASELINE config 1, SURVEY row a23:
    DetHubbard  : DetModelGC<2, double, false>           /root/reference/src/dethubbard.{h,cpp}
Literal where it matters for parity: real dgesvd UdV per spin species, the dense B = diag(e^{+-alpha s}) * proptmat
built as a product over slices (dethubbard.cpp:823-849), numerical inverses arma::inv(B) (dethubbard.h:299-337),
rank-1 Sherman-Morrison flips in random site order with replacement (:141-172, :858-906), the sweep skeleton of
detmodel.h:1266-1478.  Pinned by tests/golden/hub_*.npz produced by the real reference (oracle/ref_build, target
`hubbard`; tests/test_oracle_vs_golden.py).
"""
from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla

from dsfmt_oracle import RngWrapper


@dataclass
class HubbardParams:
    """ModelParams<DetHubbard> (src/dethubbardparams.h:28-50)"""
    L: int = 4
    d: int = 2
    beta: float = 2.0
    dtau: float = 0.1
    s: int = 10
    t: float = 1.0
    U: float = 4.0
    mu: float = 0.0
    checkerboard: bool = False
    rngSeed: int = 1020304050
    simindex: int = 0


class UdVReal:
    def __init__(self, U, d, V_t):
        self.U, self.d, self.V_t = U, d, V_t


def udv(M):
    """udvDecompose<num> (src/udv.h:68-102): arma::svd "std" -> dgesvd; M = U diag(d) V_t^T"""
    U, d, Vh = sla.svd(M, full_matrices=True, lapack_driver="gesvd", check_finite=False)
    return UdVReal(U, d, Vh.T)


class DetHubbardOracle:
    UP, DN = 0, 1

    def __init__(self, pars: HubbardParams, rng: RngWrapper = None):
        p = self.pars = pars
        if p.d != 2:
            raise ValueError("oracle restates the 2d lattice only")
        self.rng = rng if rng is not None else RngWrapper(p.rngSeed, p.simindex + 1)       # src/detqmc.h:181
        # updateTemperatureParameters (src/detmodelparams.h:68-122)
        self.m = int(round(p.beta / p.dtau))
        self.beta = self.m * p.dtau
        self.s = p.s
        while self.m <= self.s:
            self.s -= 1
        self.n = -(-self.m // self.s)
        self.L, self.N = p.L, p.L ** p.d
        self.alpha = np.arccosh(np.exp(p.dtau * p.U * 0.5))                                 # dethubbard.cpp:55
        N, L = self.N, self.L
        # PeriodicCubicLatticeNearestNeighbors (src/neighbortable.h): XPLUS, XMINUS, YPLUS, YMINUS
        self.neigh = np.zeros((4, N), dtype=int)
        for site in range(N):
            x, y = site % L, site // L
            self.neigh[0, site] = y * L + (x + 1) % L
            self.neigh[1, site] = y * L + (x - 1) % L
            self.neigh[2, site] = ((y + 1) % L) * L + x
            self.neigh[3, site] = ((y - 1) % L) * L + x
        self.auxfield = np.zeros((N, self.m + 1))
        for k in range(1, self.m + 1):                                                      # setupRandomAuxfield, :690-700
            for site in range(N):
                self.auxfield[site, k] = +1.0 if self.rng.rand01() <= 0.5 else -1.0
        self.proptmat = self._proptmat_checkerboard() if p.checkerboard else self._proptmat_direct()
        self.performedSweeps = 0
        self.setupUdVStorage_and_calculateGreen()

    # ---- hopping propagator ----
    def _proptmat_direct(self):
        """:702-714 + computePropagator (src/detmodel.cpp:25-33)"""
        N = self.N
        tmat = -self.pars.mu * np.eye(N)
        for site in range(N):
            for dirn in range(4):
                tmat[self.neigh[dirn, site], site] -= self.pars.t
        w, v = np.linalg.eigh(tmat)
        return (v * np.exp(-self.pars.dtau * w)[None, :]) @ v.T

    def _proptmat_checkerboard(self):
        """:717-770 (dos Santos 2003): ordered product form, no chemical potential"""
        N, L = self.N, self.L
        kxa, kxb, kya, kyb = (np.zeros((N, N)) for _ in range(4))
        for y in range(L):
            for x in range(0, L, 2):
                a = y * L + x
                na = self.neigh[0, a]
                kxa[a, na] = kxa[na, a] = 1.0
                nb = self.neigh[0, na]
                kxb[na, nb] = kxb[nb, na] = 1.0
        for x in range(L):
            for y in range(0, L, 2):
                a = y * L + x
                na = self.neigh[2, a]
                kya[a, na] = kya[na, a] = 1.0
                nb = self.neigh[2, na]
                kyb[na, nb] = kyb[nb, na] = 1.0
        ch, sh = np.cosh(self.pars.dtau * self.pars.t), np.sinh(self.pars.dtau * self.pars.t)
        eye = np.eye(N)
        return (ch ** 4 * eye + ch ** 3 * sh * (kxa + kxb + kya + kyb)
                + ch ** 2 * sh ** 2 * (kxa @ kxb + kxa @ kya + kxb @ kya + kxa @ kyb + kxb @ kyb + kya @ kyb)
                + ch * sh ** 3 * (kxa @ kxb @ kya + kxa @ kxb @ kyb + kxa @ kya @ kyb + kxb @ kya @ kyb)
                + sh ** 4 * (kxa @ kxb @ kya @ kyb))

    # ---- B matrices (:772-800, dethubbard.h:279-337) ----
    def computeBmat(self, k2, k1, gc):
        if k2 == k1:
            return np.eye(self.N)
        sign = +1.0 if gc == self.UP else -1.0
        one = lambda k: np.exp(sign * self.alpha * self.auxfield[:, k])[:, None] * self.proptmat
        B = one(k2)
        for k in range(k2 - 1, k1, -1):
            B = B @ one(k)
        return B

    def leftMultiplyBmat(self, gc, A, k2, k1):
        return self.computeBmat(k2, k1, gc) @ A

    def rightMultiplyBmat(self, gc, A, k2, k1):
        return A @ self.computeBmat(k2, k1, gc)

    def leftMultiplyBmatInv(self, gc, A, k2, k1):
        return np.linalg.inv(self.computeBmat(k2, k1, gc)) @ A

    def rightMultiplyBmatInv(self, gc, A, k2, k1):
        return A @ np.linalg.inv(self.computeBmat(k2, k1, gc))

    # ---- stabilised Green's functions: src/detmodel.h:680-860, 956-1163 for both sectors ----
    @staticmethod
    def _greenFromUdV(L_, R_):
        tmp = udv(R_.U.T @ L_.V_t + (R_.d[:, None] * (R_.V_t.T @ L_.U)) * L_.d[None, :])
        return ((L_.V_t @ tmp.V_t) * (1.0 / tmp.d)[None, :]) @ (R_.U @ tmp.U).T

    @staticmethod
    def _greenFromEye(R_):
        tmp = udv(R_.U.T @ R_.V_t + np.diag(R_.d))
        return ((R_.V_t @ tmp.V_t) * (1.0 / tmp.d)[None, :]) @ (R_.U @ tmp.U).T

    def setupUdVStorage_and_calculateGreen(self):
        n, s, m, N = self.n, self.s, self.m, self.N
        self.storage = [[None] * (n + 1) for _ in range(2)]
        self.g = [None, None]
        for gc in (0, 1):
            st = self.storage[gc]
            st[0] = UdVReal(np.eye(N), np.ones(N), np.eye(N))
            st[1] = udv(self.leftMultiplyBmat(gc, np.eye(N), s, 0))
            for l in range(1, n):
                k_l, k_lp1 = s * l, (s * (l + 1) if l < n - 1 else m)
                nxt = udv(self.leftMultiplyBmat(gc, st[l].U, k_lp1, k_l) * st[l].d[None, :])
                nxt.V_t = st[l].V_t @ nxt.V_t
                st[l + 1] = nxt
            self.g[gc] = self._greenFromEye(st[n])
        self.currentTimeslice = m
        self.lastSweepDir = +1

    def advanceDownGreen(self, l):
        n, s, m, N = self.n, self.s, self.m, self.N
        k_l, k_lm1 = (s * l if l < n else m), s * (l - 1)
        for gc in (0, 1):
            st = self.storage[gc]
            if l < n:
                UL = udv(st[l].d[:, None] * self.rightMultiplyBmat(gc, st[l].V_t.T, k_l, k_lm1))
                UL.U = st[l].U @ UL.U
            else:
                UL = udv(self.rightMultiplyBmat(gc, np.eye(N), k_l, k_lm1))
            self.g[gc] = self._greenFromUdV(UL, st[l - 1]) if l - 1 > 0 else self._greenFromEye(UL)
            st[l - 1] = UL
        self.currentTimeslice = k_lm1

    def advanceUpGreen(self, l):
        n, s, m = self.n, self.s, self.m
        k_l, k_lp1 = s * l, (s * (l + 1) if l < n - 1 else m)
        for gc in (0, 1):
            st = self.storage[gc]
            tmp = udv(self.leftMultiplyBmat(gc, st[l].U, k_lp1, k_l) * st[l].d[None, :])
            tmp.V_t = st[l].V_t @ tmp.V_t
            self.g[gc] = self._greenFromUdV(st[l + 1], tmp) if k_lp1 != m else self._greenFromEye(tmp)
            st[l + 1] = tmp
        self.currentTimeslice = k_lp1

    def wrapDownGreen(self, k):
        for gc in (0, 1):
            self.g[gc] = self.leftMultiplyBmatInv(gc, self.rightMultiplyBmat(gc, self.g[gc], k, k - 1), k, k - 1)
        self.currentTimeslice = k - 1

    def wrapUpGreen(self, k):
        for gc in (0, 1):
            self.g[gc] = self.leftMultiplyBmat(gc, self.rightMultiplyBmatInv(gc, self.g[gc], k + 1, k), k + 1, k)
        self.currentTimeslice = k + 1

    # ---- local updates (:141-172, :858-906) ----
    def updateInSlice(self, k):
        N = self.N
        for _ in range(N):
            site = self.rng.randInt(0, N - 1)
            a = self.auxfield[site, k]
            expUp, expDn = np.exp(-2.0 * self.alpha * a), np.exp(2.0 * self.alpha * a)
            ratio = (1.0 + (expUp - 1.0) * (1.0 - self.g[0][site, site])) * (1.0 + (expDn - 1.0) * (1.0 - self.g[1][site, site]))
            if ratio > 1.0 or self.rng.rand01() < ratio:
                for gc, delta in ((0, expUp - 1.0), (1, expDn - 1.0)):
                    g = self.g[gc]
                    omg = np.eye(N) - g
                    factor = delta / (1.0 + delta * omg[site, site])
                    self.g[gc] = g - np.outer(g[:, site], omg[site, :]) * factor
                self.auxfield[site, k] = -a

    # ---- measurements (:501-577, :637-649) ----
    def initMeasurements(self):
        self.sums = dict(iiUp=0.0, iiDn=0.0, nUp=0.0, nDn=0.0, iiUpDn=0.0)
        self.zcorr = np.zeros(self.N)

    def measure(self):
        gU, gD, N = self.g[0], self.g[1], self.N
        for site in range(N):
            self.sums["iiUp"] += gU[site, site]
            self.sums["iiDn"] += gD[site, site]
            self.sums["iiUpDn"] += gU[site, site] * gD[site, site]
            for dirn in range(4):
                nb = self.neigh[dirn, site]
                self.sums["nUp"] += gU[site, nb]
                self.sums["nDn"] += gD[site, nb]
        u0, d0 = gU[0, 0], gD[0, 0]
        self.zcorr[0] += -2.0 * u0 * d0 + u0 + d0
        for j in range(1, N):
            self.zcorr[j] += u0 * gU[j, j] - u0 * gD[j, j] + d0 * gD[j, j] - d0 * gU[j, j] - gU[0, j] ** 2 - gD[0, j] ** 2

    def finishMeasurements(self):
        N, m, p, S = self.N, self.m, self.pars, self.sums
        o = {}
        o["occUp"] = 1.0 - S["iiUp"] / (N * m)
        o["occDn"] = 1.0 - S["iiDn"] / (N * m)
        o["occTotal"] = o["occUp"] + o["occDn"]
        o["occDouble"] = 1.0 + (S["iiUpDn"] - S["iiUp"] - S["iiDn"]) / (N * m)
        o["localMoment"] = o["occTotal"] - 2 * o["occDouble"]
        o["ePotential"] = p.U * o["occDouble"]
        o["eKinetic"] = (p.t / (N * m)) * (S["nUp"] + S["nDn"]) - p.mu * o["occTotal"]
        o["eTotal"] = o["eKinetic"] + o["ePotential"]
        self.obs = o
        self.zcorr = self.zcorr / m

    # ---- sweeps (src/detmodel.h:1266-1478) ----
    def _sweep(self, takeMeasurements):
        n, s, m = self.n, self.s, self.m
        if takeMeasurements:
            self.initMeasurements()

        def upd(k):
            self.updateInSlice(k)
            if takeMeasurements:
                self.measure()

        if self.lastSweepDir == +1:
            for k in range(m, (n - 1) * s, -1):
                upd(k)
                self.wrapDownGreen(k)
            for l in range(n - 1, 0, -1):
                self.advanceDownGreen(l + 1)
                for k in range(l * s, (l - 1) * s, -1):
                    upd(k)
                    self.wrapDownGreen(k)
            self.advanceDownGreen(1)
            self.lastSweepDir = -1
        else:
            N = self.N
            for gc in (0, 1):
                self.storage[gc][0] = UdVReal(np.eye(N), np.ones(N), np.eye(N))
            for l in range(0, n - 1):
                for k in range(l * s + 1, (l + 1) * s + 1):
                    self.wrapUpGreen(k - 1)
                    upd(k)
                self.advanceUpGreen(l)
            for k in range((n - 1) * s + 1, m + 1):
                self.wrapUpGreen(k - 1)
                upd(k)
            self.advanceUpGreen(n - 1)
            self.lastSweepDir = +1
        self.performedSweeps += 1
        if takeMeasurements:
            self.finishMeasurements()

    def sweepThermalization(self):
        self._sweep(False)

    def sweep(self, takeMeasurements=False):
        self._sweep(takeMeasurements)
