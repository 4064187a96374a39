import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # oracle is test infrastructure: tests may import it
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["params"] = json.loads(str(d.pop("params_json")))
    return d


def oracle_params(args):
    """harness key=value args (oracle/make_golden.py) -> oracle SDWParams."""
    from detsdw_oracle import SDWParams
    a = dict(args)
    kw = {}
    for k in ("opdim", "L", "s", "delaySteps", "globalUpdateInterval"):
        if k in a:
            kw[k] = int(a[k])
    for k in ("beta", "dtau", "r", "c", "u", "txhor", "txver", "tyhor", "tyver", "mu", "mux", "muy", "accRatio", "cdwU"):
        if k in a:
            kw[k] = float(a[k])
    if "lambda" in a:
        kw["lambda_"] = float(a["lambda"])
    if "bc" in a:
        kw["bc"] = a["bc"]
    kw["weakZflux"] = bool(int(a.get("weakZflux", 0)))
    kw["globalShift"] = bool(int(a.get("globalShift", 0)))
    kw["checkerboard"] = bool(int(a.get("checkerboard", 1)))
    kw["wolffClusterUpdate"] = bool(int(a.get("wolffClusterUpdate", 0)))
    kw["wolffClusterShiftUpdate"] = bool(int(a.get("wolffClusterShiftUpdate", 0)))
    kw["repeatWolffPerSweep"] = int(a.get("repeatWolffPerSweep", 1))
    kw["turnoffFermionMeasurements"] = not bool(int(a.get("fermionMeas", 0)))
    kw["spinProposalMethod"] = a.get("spinProposalMethod", "box")
    kw["adaptScaleVariance"] = bool(int(a.get("adaptScaleVariance", 0)))
    kw["repeatUpdateInSlice"] = int(a.get("repeatUpdateInSlice", 1))
    kw["rngSeed"] = int(a.get("rngSeed", 1020304050))
    kw["simindex"] = int(a.get("simindex", 0))
    return SDWParams(**kw)


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def relerr_blocks(a, b, bs=16, floor=1e-6):
    """Block-wise relative error of two matrices: the largest, over all bs x bs blocks, of max|a - b| in the block divided by the
    block's own scale max|b| (not below floor x the global scale).  relerr() above is relative to the largest entry of the whole
    matrix, which for a Green's function sits on the diagonal: an error confined to the small far-off-diagonal blocks would pass it."""
    a = np.asarray(a)
    b = np.asarray(b)
    n0, n1 = b.shape
    p0, p1 = (-n0) % bs, (-n1) % bs
    d = np.pad(np.abs(a - b), ((0, p0), (0, p1)))
    s = np.pad(np.abs(b), ((0, p0), (0, p1)))
    D = d.reshape(d.shape[0] // bs, bs, d.shape[1] // bs, bs).max(axis=(1, 3))
    S = s.reshape(s.shape[0] // bs, bs, s.shape[1] // bs, bs).max(axis=(1, 3))
    return float(np.max(D / np.maximum(S, floor * max(np.max(S), 1e-300))))
