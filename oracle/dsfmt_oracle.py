"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product package.

CPU restatement of the random-number source whose *consumption order defines the Markov
chain* of the reference ("RNG wrapper + dSFMT"):

  * dSFMT-19937 (Saito & Matsumoto, "A PRNG specialized in double precision floating point
    numbers using an affine transition", 2009; third-party BSD code vendored by the reference
    under src/dsfmt/, version 2.x, parameter set dSFMT2-19937:117-19:ffafffffffb3f-ffdfffc90fffd).
    Restated here from the published recursion; constants as in src/dsfmt/dSFMT-params19937.h:6-17
    and src/dsfmt/dSFMT-params.h:33-35; seeding as dsfmt_chk_init_gen_rand (src/dsfmt/dSFMT.c:625-646),
    initial_mask (:428-436), period_certification (:442-467), block refill dsfmt_gen_rand_all
    (:509-525), output map dsfmt_genrand_open_open (src/dsfmt/dSFMT.h:341-355).
  * RngWrapper seed mangling and rand01/randRange/randInt (src/rngwrapper.cpp:31-49,
    src/rngwrapper.h:54-68).

Pinned by tests/golden/rng.npz, which holds numbers drawn from the real reference build
(oracle/ref_build, mode=rng).
"""
import numpy as np

_N = 191
_POS1 = 117
_SL1 = 19
_SR = 12
_MSK1 = 0x000FFAFFFFFFFB3F
_MSK2 = 0x000FFDFFFC90FFFD
_FIX1 = 0x90014964B32F4329
_FIX2 = 0x3B8D12AC548A7C7A
_PCV1 = 0x3D84E1AC0DC82880
_PCV2 = 0x0000000000000001
_LOW_MASK = 0x000FFFFFFFFFFFFF
_HIGH_CONST = 0x3FF0000000000000
_M64 = 0xFFFFFFFFFFFFFFFF
_N64 = 2 * _N


class DSFMT19937:
    """State: 191 128-bit words (as pairs of python ints) + the 128-bit 'lung'."""

    def __init__(self, seed):
        # dsfmt_chk_init_gen_rand: 32-bit LCG fill of (N+1)*4 words
        n32 = (_N + 1) * 4
        ps = [0] * n32
        ps[0] = seed & 0xFFFFFFFF
        for i in range(1, n32):
            prev = ps[i - 1]
            ps[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF
        # little endian: 64-bit word j = ps[2j] | ps[2j+1] << 32
        st = [ps[2 * j] | (ps[2 * j + 1] << 32) for j in range((_N + 1) * 2)]
        # initial_mask on the first 2N 64-bit words
        for j in range(2 * _N):
            st[j] = (st[j] & _LOW_MASK) | _HIGH_CONST
        # period_certification on the lung
        t0 = st[2 * _N] ^ _FIX1
        t1 = st[2 * _N + 1] ^ _FIX2
        inner = (t0 & _PCV1) ^ (t1 & _PCV2)
        i = 32
        while i > 0:
            inner ^= inner >> i
            i >>= 1
        if (inner & 1) != 1:
            st[2 * _N + 1] ^= 1   # PCV2 & 1 == 1 branch
        self.st = st
        self.idx = _N64
        self._buf = np.empty(_N64, dtype=np.float64)

    def _gen_rand_all(self):
        st = self.st
        L0 = st[2 * _N]
        L1 = st[2 * _N + 1]
        for i in range(_N):
            a0 = st[2 * i]
            a1 = st[2 * i + 1]
            bi = i + _POS1
            if bi >= _N:
                bi -= _N
            b0 = st[2 * bi]
            b1 = st[2 * bi + 1]
            nL0 = ((a0 << _SL1) & _M64) ^ (L1 >> 32) ^ ((L1 << 32) & _M64) ^ b0
            nL1 = ((a1 << _SL1) & _M64) ^ (L0 >> 32) ^ ((L0 << 32) & _M64) ^ b1
            L0, L1 = nL0, nL1
            st[2 * i] = (L0 >> _SR) ^ (L0 & _MSK1) ^ a0
            st[2 * i + 1] = (L1 >> _SR) ^ (L1 & _MSK2) ^ a1
        st[2 * _N] = L0
        st[2 * _N + 1] = L1
        # dsfmt_genrand_open_open: bits | 1, as double in [1,2), minus 1
        bits = np.array(st[:_N64], dtype=np.uint64) | np.uint64(1)
        self._buf = bits.view(np.float64) - 1.0

    def genrand_open_open(self):
        if self.idx >= _N64:
            self._gen_rand_all()
            self.idx = 0
        v = self._buf[self.idx]
        self.idx += 1
        return float(v)


class RngWrapper:
    """src/rngwrapper.h:42-68, src/rngwrapper.cpp:31-49."""

    def __init__(self, seed=0, processIndex=0):
        u32 = 0xFFFFFFFF
        a = (seed * 181) & u32
        b = (((processIndex - 83) & u32) * 359) & u32
        self.seed = seed
        self.processIndex = processIndex
        self.mySeed = ((a * b) & u32) % 104729
        self.gen = DSFMT19937(self.mySeed)
        self.count = 0

    def rand01(self):
        self.count += 1
        return self.gen.genrand_open_open()

    def randRange(self, low, high):
        return low + (high - low) * self.rand01()

    def randInt(self, low, high):
        return low + int((high - low + 1.0) * self.rand01())

    def randPointOnSphere(self):
        """rngwrapper.h:70-80"""
        import math
        phi = self.randRange(0.0, 2.0 * math.pi)
        costheta = self.randRange(-1.0, 1.0)
        sintheta = math.sqrt(1.0 - costheta * costheta)
        return (math.cos(phi) * sintheta, math.sin(phi) * sintheta, costheta)

    def randPointOnCircle(self):
        """rngwrapper.h:82-87"""
        import math
        phi = self.randRange(0.0, 2.0 * math.pi)
        return (math.cos(phi), math.sin(phi))
