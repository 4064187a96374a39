"""Developer probe: QR decompositions only (64 chains, n_g = 512) -- run under rocprofv3 --kernel-trace to see the time per
launch shape of the panel / block-reflector kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from detqmc_amd import KernelContext
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = KernelContext(2, 16, 20, 10, 0.1, delaySteps=8, stabilisation="qr", nchains=nch)
n = ctx.ng
rng = np.random.default_rng(1)
M = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * np.logspace(3, -3, n)[None, :]
for i in range(reps):
    ctx.udvDecompose(M)
print("done")
