#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Picks the rngSeed of the cdwU != 0 fixtures.

With cdwU != 0 one proposal in four of the cdwl pass draws the value the site already has.  Its acceptance probability is 1 in exact
arithmetic (a uniform is drawn, the null update accepted); in the reference's floating point it is 1 or 1 + 2^-52 depending on the last
bit of a determinant, and at 1 + 2^-52 no uniform is drawn (src/detsdwopdim.cpp:3110-3113) -- from there on the reference's own chain
depends on its BLAS.  The oracle and the HIP path take the exact-arithmetic branch (oracle/detsdw_oracle.py, updateInSlice_delayed).
This script runs the reference harness over a range of seeds and prints the first seed per case on which the reference took that
branch at every null proposal of the fixture's trajectory, i.e. on which oracle and reference walk the same chain.

    python oracle/find_cdw_seeds.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_params, relerr          # noqa: E402
from detsdw_oracle import DetSDWOracle              # noqa: E402
from make_golden import CASES, REFDIR               # noqa: E402


def load(td):
    out = {}
    for line in open(os.path.join(td, "manifest.txt")):
        parts = line.split()
        nm, dt, shape = parts[0], parts[1], tuple(int(x) for x in parts[2:])
        raw = np.fromfile(os.path.join(td, nm + ".bin"), dtype=np.complex128 if dt == "c16" else np.float64)
        out[nm] = raw.reshape(shape, order="F")
    return out


def agrees(args, g):
    o = DetSDWOracle(oracle_params(args))
    m = o.m
    if int(args.get("sliceTrace", 1)):
        o.updateInSliceThermalization(m)
        if not (np.array_equal(o.phi[m], g["slice_phi_m"]) and np.array_equal(o.cdwl[m], g["slice_cdwl_m"].reshape(-1))):
            return False
        if relerr(o.g, g["slice_g"]) > 1e-10:
            return False
        if int(args["sliceTrace"]) == 2:
            return True
        o2 = DetSDWOracle(oracle_params(args))
    else:
        o2 = o
    i = 1
    while f"sweep{i}_phi" in g:
        o2.sweepThermalization()
        if not np.array_equal(o2.cdwl[1:], g[f"sweep{i}_cdwl"].T[1:]):
            return False
        if not np.array_equal(o2.phi[1:], np.transpose(g[f"sweep{i}_phi"], (2, 0, 1))[1:]):
            return False
        i += 1
    i = 1
    while f"meas{i}_phi" in g:
        o2.sweep(True)
        if not np.array_equal(o2.cdwl[1:], g[f"meas{i}_cdwl"].T[1:]):
            return False
        i += 1
    return np.array_equal(np.array([o2.rng.rand01() for _ in range(4)]), g["rng_next"]) if "rng_next" in g else True


def main():
    names = sys.argv[1:] or [n for n in CASES if "_cdw" in n]
    for name in names:
        spec = CASES[name]
        found = None
        for seed in range(1000, 1400):
            args = dict(spec["args"], rngSeed=seed)
            exe = os.path.join(REFDIR, "ref_harness_o%d" % args["opdim"])
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([exe, td] + [f"{k}={v}" for k, v in args.items()], check=True, stdout=subprocess.DEVNULL,
                               env=dict(os.environ, MKL_NUM_THREADS="1"))
                g = load(td)
            if agrees(args, g):
                found = seed
                break
        print(name, "rngSeed =", found, flush=True)


if __name__ == "__main__":
    main()
