"""QR ("UDT") decomposition at sizes beyond the register-resident panel: n_g = 784, 1296, 2304 (O(3) L = 24, BASELINE
config 5) and 4096.  M = Q diag(d) T with Q unitary; reconstruction and unitarity against numpy."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from detqmc_amd import KernelContext

cases = [(3, 14), (3, 18), (3, 24), (2, 32)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for opdim, L in cases:
    t0 = time.time()
    ctx = KernelContext(opdim, L, 20, 10, 0.1, delaySteps=8, stabilisation="qr")
    n = ctx.ng
    rng = np.random.default_rng(1)
    W = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    M = W * np.logspace(4, -4, n)[None, :]
    ctx.udvDecompose(M)                                  # warm-up (module load)
    t1 = time.time()
    U, d, Vt, _ = ctx.udvDecompose(M)
    dt = time.time() - t1
    err = np.max(np.abs((U * d[None, :]) @ Vt.conj().T - M)) / np.max(np.abs(M))
    ortho = np.max(np.abs(U.conj().T @ U - np.eye(n)))
    condT = np.linalg.cond(Vt)
    print("n_g %5d  decompose incl. transfers %.3fs  reconstruction %.2e  unitarity %.2e  cond(T) %.1f" % (n, dt, err, ortho, condT), flush=True)
    assert err < 1e-11 and ortho < 1e-11
    ctx.close()
print("ok")
