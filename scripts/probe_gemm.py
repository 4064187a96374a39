"""Developer probe: n_g^3 product for the four op combinations (64 chains, n_g = 512), device time from the profile counters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from detqmc_amd import KernelContext
ctx = KernelContext(2, 16, 20, 10, 0.1, delaySteps=8, stabilisation="qr", nchains=64)
n = ctx.ng
rng = np.random.default_rng(1)
A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
for opA in (0, 1):
    for opB in (0, 1):
        ctx.gemm(opA, opB, A, B)
        ctx.profile_enable(True)
        for i in range(5):
            Cm = ctx.gemm(opA, opB, A, B)
        pr = ctx.profile_read()
        ctx.profile_enable(False)
        ref = (A.conj().T if opA else A) @ (B.conj().T if opB else B)
        print("opA %d opB %d: %.0f us per launch (64 chains), %.1f TFLOP/s, err %.1e" % (
            opA, opB, 1e3 * pr["gemm"][0] / pr["gemm"][1], 8 * n ** 3 * 64 / (pr["gemm"][0] / pr["gemm"][1] * 1e-3) / 1e12,
            np.max(np.abs(Cm - ref)) / np.max(np.abs(ref))), flush=True)
