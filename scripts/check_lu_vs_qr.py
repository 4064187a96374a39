"""Developer check: the Markov chains of a batch with the LU-based Green's function (default) and with the Householder route
(SDWParams.greenVariant = 1) must be the same chains.  Run twice and compare the records:

    python scripts/check_lu_vs_qr.py 16 10 30 8 > lu.txt;  python scripts/check_lu_vs_qr.py 16 10 30 8 qr > qr.txt
    python scripts/check_lu_vs_qr.py --compare lu.txt qr.txt

Per sweep and chain: SHA-256 of the field, accepted / attempted global shifts (their decision uses log det from diag U resp. diag R)
and G rounded to 1e-9 (hashed)."""
import sys, os, hashlib, dataclasses, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "--compare":
    a = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")]
    b = [json.loads(l) for l in open(sys.argv[3]) if l.startswith("{")]
    assert len(a) == len(b) and len(a) > 0
    worst = 0.0
    for x, y in zip(a, b):
        assert x["phi"] == y["phi"], "fields differ at sweep %d" % x["sweep"]
        assert x["shifts"] == y["shifts"], "global-move decisions differ at sweep %d" % x["sweep"]
        worst = max(worst, max(abs(p - q) / max(abs(q), 1e-300) for p, q in zip(x["gsum"], y["gsum"])))
    assert worst < 1e-9, worst
    print("same chains over %d sweeps x %d chains; checksum of G agrees to %.1e" % (len(a), len(a[0]["phi"]), worst))
    sys.exit(0)

from detqmc_amd import DetSDWBatch, SDWParams
L, beta, nsw, B = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
green_variant = 1 if (len(sys.argv) > 5 and sys.argv[5] == "qr") else 0
p0 = SDWParams(opdim=2, L=L, beta=beta, s=10, delaySteps=32, stabilisation="qr", globalShift=True, globalUpdateInterval=3,
               greenVariant=green_variant)
batch = DetSDWBatch([dataclasses.replace(p0, simindex=b, r=p0.r + 0.02 * b) for b in range(B)], sub_batches=1)
for sw in range(nsw):
    batch.sweepThermalization()
    rec = {"sweep": sw + 1, "phi": [], "shifts": [], "gsum": []}
    for b in range(B):
        c = batch.chain(b)
        rec["phi"].append(hashlib.sha256(np.ascontiguousarray(c.phi).tobytes()).hexdigest())
        rec["shifts"].append([c.info.acceptedGlobalShifts, c.info.attemptedGlobalShifts])
        g = c.g
        rec["gsum"].append(float(np.sum(np.abs(g) ** 2)))
    print(json.dumps(rec), flush=True)
batch.close()
