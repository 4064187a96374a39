import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from detqmc_amd import KernelContext
t0 = time.time()
ctx = KernelContext(3, 24, 20, 10, 0.1, delaySteps=8)
n = ctx.ng
print("n_g", n, "create %.1fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(1)
W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
M = W * np.logspace(4, -4, n)[None, :]
t0 = time.time()
U, d, Vt, sweeps = ctx.udvDecompose(M)
print("decompose %.1fs, jacobi sweeps %s" % (time.time() - t0, sweeps), flush=True)
err = np.max(np.abs((U * d[None, :]) @ Vt.conj().T - M)) / np.max(np.abs(M))
ortho = np.max(np.abs(U.conj().T @ U - np.eye(n)))
print("reconstruction %.2e  orthogonality %.2e  d sorted %s" % (err, ortho, bool(np.all(np.diff(d) <= 0))))
assert err < 1e-11 and ortho < 1e-11
