"""Developer probe: per-family device time and Jacobi statistics of a few sweeps."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from detqmc_amd import DetSDW, SDWParams

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
beta = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stab = sys.argv[4] if len(sys.argv) > 4 else "svd"
t0 = time.time()
rep = DetSDW(SDWParams(opdim=2, L=L, beta=beta, s=10, delaySteps=16, stabilisation=stab))
print("init %.3f s" % (time.time() - t0), flush=True)
ctx = rep.kernel_context
for i in range(2):
    rep.sweepThermalization()
ctx.profile_enable(True)
t0 = time.time()
for i in range(nsw):
    rep.sweepThermalization()
ctx.synchronize()
dt = time.time() - t0
pr = ctx.profile_read()
print("sweeps/s %.3f  (%.1f ms/sweep)" % (nsw / dt, 1e3 * dt / nsw))
for k, v in pr.items():
    print(k, v)
print("acc", rep.info.lastAccRatioLocal_phi, "phiDelta", rep.info.phiDelta)
