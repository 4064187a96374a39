"""Developer check: SVD-mode and QR-mode replicas started from the same seed must walk the same Markov chain and agree in
G to ~1e-10 (here at sizes / temperatures without a reference fixture, e.g. BASELINE config 4: L = 16, beta = 20)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from detqmc_amd import DetSDW, SDWParams

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
beta = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 3
opdim = int(sys.argv[4]) if len(sys.argv) > 4 else 2
flux = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False          # BASELINE config 5: O(3) L = 24 with magnetic flux
kw = dict(opdim=opdim, L=L, beta=beta, s=10, delaySteps=16, globalShift=True, globalUpdateInterval=2, weakZflux=flux)
a = DetSDW(SDWParams(stabilisation="svd", **kw))
b = DetSDW(SDWParams(stabilisation="qr", **kw))
ga, gb = a.g, b.g
print("init: G rel diff %.2e, logdet diff %.2e" % (np.max(np.abs(ga - gb)) / np.max(np.abs(ga)),
      abs(np.sum(np.log(a.g_inv_sv)) - np.sum(np.log(b.g_inv_sv)))), flush=True)
for i in range(nsw):
    t0 = time.time(); a.sweepThermalization(); ta = time.time() - t0
    t0 = time.time(); b.sweepThermalization(); tb = time.time() - t0
    ga, gb = a.g, b.g
    same = np.array_equal(a.phi, b.phi)
    print("sweep %d: same field %s, G rel diff %.2e, acc %.3f, svd %.2fs qr %.2fs, shifts %d/%d" % (
        i + 1, same, np.max(np.abs(ga - gb)) / np.max(np.abs(ga)), a.info.lastAccRatioLocal_phi, ta, tb,
        a.info.acceptedGlobalShifts, a.info.attemptedGlobalShifts), flush=True)
    assert same
    assert np.max(np.abs(ga - gb)) / np.max(np.abs(ga)) < 1e-9
print("ok")
