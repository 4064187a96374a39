"""Developer probe: cost of measurement sweeps (bosonic + fermionic observables on the device) vs plain sweeps."""
import sys, time, dataclasses
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detqmc_amd import DetSDWBatch, SDWParams

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
p0 = SDWParams(opdim=2, L=L, beta=10.0, s=10, delaySteps=32, stabilisation="qr", fermionMeasurements=True)
batch = DetSDWBatch([dataclasses.replace(p0, simindex=b) for b in range(B)])
ctx = batch.kernel_context
batch.sweepThermalization(); batch.sweepThermalization()
for label, fn in (("sweepThermalization", batch.sweepThermalization), ("sweep(False)", lambda: batch.sweep(False)),
                  ("sweep(True) with fermionic measurements", lambda: batch.sweep(True))):
    ctx.synchronize(); t0 = time.time()
    for _ in range(2):
        fn()
    ctx.synchronize(); dt = (time.time() - t0) / 2
    print("%-42s %.1f ms per lockstep sweep of %d chains" % (label, 1e3 * dt, B), flush=True)
o = batch.chain(0).observables
print("chain 0: greenK0 %.6f greenLocal %.6f occDiffSq %.6f pairPlusMax %.3e normMeanPhi %.4f" % (o.greenK0, o.greenLocal, o.occDiffSq, o.pairPlusMax, o.normMeanPhi))
print("kOccX[:4]", batch.chain(0).observable_vector("kOccX")[:4])
