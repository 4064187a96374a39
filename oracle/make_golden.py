#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz from the REAL reference.

Runs oracle/_ref/ref_harness_o{1,2,3} (built by `make -C oracle/ref_build OPDIM=n` from the
reference sources where they lie under /root/reference) for a fixed list of parameter sets
and packs the raw dumps into small .npz fixtures.  The fixtures are data only: inputs
(parameters, seeds) and the reference's outputs (fields, Green's functions, singular values).
Run in the build container only -- the GPU box has no /root/reference and uses the committed
fixtures.

    python oracle/make_golden.py [case ...]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
OUTDIR = os.path.join(ROOT, "tests", "golden")

# name -> (harness args, keep-filter or None, subsample big matrices?)
CASES = {
    "o2_L4": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=4, dumpUdV=1)),
    "o2_L4_s7": dict(args=dict(opdim=2, L=4, beta=2.3, s=7, delaySteps=16, sweeps=3, cfgStream=1)),
    "o2_L4_flux": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=3, weakZflux=1)),
    "o2_L4_apbc": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=3, bc="apbc-xy",
                                 mux=-0.4, muy=-0.7, mu=-0.5, r=0.5, c=2.0, u=0.7)),
    "o2_L4_gshift": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=6, globalShift=1,
                                   globalUpdateInterval=2, sliceTrace=0)),
    # Wolff cluster moves (a21): single cluster update, combined cluster + shift, all three order-parameter dimensions
    "o2_L4_wolff": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=6, wolffClusterUpdate=1, globalShift=1,
                                  globalUpdateInterval=1, repeatWolffPerSweep=2, sliceTrace=0)),
    "o2_L4_wolffshift": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=6, wolffClusterShiftUpdate=1,
                                       globalUpdateInterval=2, sliceTrace=0)),
    "o1_L4_wolff": dict(args=dict(opdim=1, L=4, beta=2, s=10, delaySteps=6, sweeps=5, wolffClusterUpdate=1,
                                  globalUpdateInterval=1, sliceTrace=0)),
    "o3_L4_wolff": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=5, wolffClusterShiftUpdate=1,
                                  globalUpdateInterval=1, sliceTrace=0)),
    # sweep(takeMeasurements = true): bosonic observables
    "o2_L4_meas": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=2, measureSweeps=3, globalShift=1,
                                 globalUpdateInterval=2, sliceTrace=0)),
    "o3_L4_meas": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=1, measureSweeps=2, sliceTrace=0)),
    # measurement sweeps incl. the fermionic observables (shiftGreenSymmetric, k-space occupation, pairing, ...)
    "o2_L4_fmeas": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=1, measureSweeps=2, fermionMeas=1, sliceTrace=0)),
    "o2_L4_fmeas_apbc_flux": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=1, measureSweeps=2, fermionMeas=1,
                                            sliceTrace=0, bc="apbc-x", weakZflux=1)),
    "o3_L4_fmeas": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=1, measureSweeps=2, fermionMeas=1, sliceTrace=0)),
    "o1_L4_fmeas": dict(args=dict(opdim=1, L=4, beta=2, s=10, delaySteps=6, sweeps=1, measureSweeps=1, fermionMeas=1, sliceTrace=0,
                                  checkerboard=0)),
    # headline size, longer: 6 sweeps with global shift moves every other sweep; fields as SHA-256, G as checksums
    "o2_L16_b10_long": dict(args=dict(opdim=2, L=16, beta=10, s=10, delaySteps=16, sweeps=6, sliceTrace=0, globalShift=1,
                                      globalUpdateInterval=2),
                            drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv"),
                            subsample=32, hash_fields=True),
    "o2_L6_seed": dict(args=dict(opdim=2, L=6, beta=3, s=10, delaySteps=8, sweeps=2, rngSeed=5555, simindex=3)),
    # checkerboard=false (CB_NONE): dense B = e^{-dtau V} e^{-dtau K}, inverse by arma::inv
    "o2_L4_dense": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=3, checkerboard=0)),
    "o2_L4_dense_flux": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=2, checkerboard=0, weakZflux=1,
                                      bc="apbc-x", mux=-0.3, muy=-0.6)),
    # round 3: cdwU != 0 -- the discrete four-valued field l_i(tau) next to phi (setupRandomField :1099-1113, evMatrix :3187-3229,
    # proposeNewCDWl :4173-4182, the second updateInSlice pass :2474-2485)
    # (the seeds: oracle/find_cdw_seeds.py -- trajectories on which the reference's last-bit decision at null cdwl proposals is the
    #  exact-arithmetic one; for O(3) that decision is a coin flip per null proposal, so the O(3) fixture stops after one slice)
    "o2_L4_cdw": dict(args=dict(rngSeed=1021, opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=2, cdwU=0.5, sliceTrace=0)),
    "o1_L4_cdw": dict(args=dict(rngSeed=1006, opdim=1, L=4, beta=2, s=10, delaySteps=6, sweeps=2, cdwU=0.4, sliceTrace=0)),
    "o2_L4_cdw_slice": dict(args=dict(rngSeed=1000, opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=0, cdwU=0.5, sliceTrace=2)),
    "o3_L4_cdw_slice": dict(args=dict(rngSeed=1003, opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=0, cdwU=0.7, sliceTrace=2)),
    "o3_L4_cdw": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=1, cdwU=0.7, sliceTrace=0)),
    "o2_L4_cdw_gshift": dict(args=dict(rngSeed=1018, opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=2, cdwU=0.5, globalShift=1, globalUpdateInterval=1,
                                       sliceTrace=0, weakZflux=1)),
    "o2_L4_cdw_dense": dict(args=dict(rngSeed=1007, opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=1, cdwU=0.5, checkerboard=0, sliceTrace=0)),
    # round 4: flux / antiperiodic boundaries on an 8 x 8 lattice -- 16 plaquettes per subgroup, Landau-gauge phases e^{-+2 pi i zmag y}
    # for y = 0 .. 7, the boundary-crossing vertical phase e^{+-2 pi i zmag L x} on 4 plaquettes per band (detsdwopdim.cpp:1598-1684),
    # APBC sign flips on both boundary rows of plaquettes (:1809-1816), mu_x != mu_y; m = 23 with s = 10 (s does not divide m)
    "o2_L8_flux": dict(args=dict(opdim=2, L=8, beta=2.3, s=10, delaySteps=12, sweeps=2, weakZflux=1, mux=-0.4, muy=-0.7, mu=-0.5),
                       drop=("bdense", "init_coshTermPhi", "init_sinhTermPhi")),
    "o2_L8_apbc_flux": dict(args=dict(opdim=2, L=8, beta=2, s=10, delaySteps=16, sweeps=2, weakZflux=1, bc="apbc-xy", r=0.5, c=2.0, u=0.7),
                            drop=("bdense", "init_coshTermPhi", "init_sinhTermPhi")),
    "o2_L8_apbc": dict(args=dict(opdim=2, L=8, beta=2, s=10, delaySteps=16, sweeps=2, bc="apbc-xy", mux=-0.4, muy=-0.7, mu=-0.5),
                       drop=("bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi")),
    "o2_L8_dense_flux": dict(args=dict(opdim=2, L=8, beta=2, s=10, delaySteps=16, sweeps=2, checkerboard=0, weakZflux=1, bc="apbc-y"),
                             drop=("bchain_", "bmult_leftinv", "bmult_rightinv", "init_coshTermPhi", "init_sinhTermPhi", "slice_g_wrapped")),
    "o2_L8_fmeas_apbc_flux": dict(args=dict(opdim=2, L=8, beta=2, s=10, delaySteps=16, sweeps=1, measureSweeps=1, fermionMeas=1,
                                            sliceTrace=0, bc="apbc-y", weakZflux=1),
                                  drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv")),
    # round 4: rotate / scale proposals of the O(3) model and their adaptation (proposeRotatedPhi / proposeScaledPhi /
    # proposeRotatedScaledPhi, detsdwopdim.cpp:3934-4170; ADAPT_ROTATE / ADAPT_SCALE :3299-3375; Box-Muller stack normaldistribution.h)
    # and repeatUpdateInSlice > 1 (:2438).  12 sweeps: the rotate and the scale running averages each reach 100 samples (m = 20 slices
    # per sweep, alternating sweeps), so angleDelta and scaleDelta move.
    "o3_L4_rotscale": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=12, spinProposalMethod="rotate_then_scale",
                                     adaptScaleVariance=1, sliceTrace=0),
                           drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv")),
    "o3_L4_rotandscale": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=6, spinProposalMethod="rotate_and_scale",
                                        adaptScaleVariance=1, sliceTrace=0),
                              drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv")),
    "o3_L6_rotscale_rep2": dict(args=dict(opdim=3, L=6, beta=1.5, s=10, delaySteps=12, sweeps=2, spinProposalMethod="rotate_then_scale",
                                          repeatUpdateInSlice=2, sliceTrace=0),
                                drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv"), subsample=4),
    "o2_L4_rep3": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=2, repeatUpdateInSlice=3, sliceTrace=0),
                       drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi", "init_udv")),
    "o1_L4": dict(args=dict(opdim=1, L=4, beta=2, s=10, delaySteps=6, sweeps=3)),
    "o3_L4": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=3)),
    # BASELINE config 2 (bring-up size): full G only at a few points
    "o2_L8_b5": dict(args=dict(opdim=2, L=8, beta=5, s=10, delaySteps=16, sweeps=2),
                     drop=("bchain_", "bdense", "bmult_leftinv", "bmult_rightinv", "slice_g_wrapped",
                           "init_coshTermPhi", "init_sinhTermPhi")),
    # BASELINE config 4 temperature (beta = 20) at a size whose matrices still fit a fixture
    "o2_L8_b20": dict(args=dict(opdim=2, L=8, beta=20, s=10, delaySteps=16, sweeps=2, sliceTrace=0),
                      drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi"), subsample=4),
    "o3_L6": dict(args=dict(opdim=3, L=6, beta=3, s=10, delaySteps=12, sweeps=2, sliceTrace=0),
                  drop=("bchain_", "bdense", "bmult_leftinv", "bmult_rightinv", "init_coshTermPhi", "init_sinhTermPhi"), subsample=4),
    # BASELINE config 4 (O(2) L = 16, beta = 20: m = 200, n = 20): 2 sweeps, a global shift move in the first; fields as
    # SHA-256, G as checksums
    "o2_L16_b20": dict(args=dict(opdim=2, L=16, beta=20, s=10, delaySteps=16, sweeps=2, sliceTrace=0, globalShift=1,
                                 globalUpdateInterval=2, setupOnly=1),
                       drop=("init_coshTermPhi", "init_sinhTermPhi", "init_udv"), subsample=32, hash_fields=True),
    # BASELINE config 5's size (O(3) L = 24, beta = 20, n_g = 2304; no flux -- the reference rejects weakZflux for opdim 3,
    # src/detsdwparams.cpp:57-60): the state after construction only (200 B-multiplies + 20 SVDs of 2304 x 2304 on the CPU)
    # round 3: plus ONE updateInSlice (k = m) and the wrap behind it -- 576 accept / reject decisions at n_g = 2304 from the reference
    "o3_L24_b20_init": dict(args=dict(opdim=3, L=24, beta=20, s=10, delaySteps=16, sweeps=0, sliceTrace=2, setupOnly=1),
                            drop=("init_coshTermPhi", "init_sinhTermPhi", "init_udv_U", "init_udv_Vt", "sweep"), subsample=64,
                            hash_fields=True, threads=4),
    # one FULL Green's function at the headline size (closes the gap the sub-sampled checksums leave)
    "o2_L16_b10_fullG": dict(args=dict(opdim=2, L=16, beta=10, s=10, delaySteps=16, sweeps=1, sliceTrace=0, setupOnly=1),
                             keep=("sweep1_g", "sweep1_phi", "init_phi", "meta"), hash_fields=True),
    # a19: the reference's immediate-update variants (updateInSlice_woodbury / _iterative, src/detsdwopdim.cpp:2493-3019)
    "o2_L4_woodbury": dict(args=dict(opdim=2, L=4, beta=2, s=10, delaySteps=6, sweeps=3, updateMethod="woodbury", sliceTrace=0),
                           keep=("meta", "sweep1_phi", "sweep2_phi", "sweep3_phi", "sweep3_g", "sweep3_phiDelta", "rng_next")),
    # (updateMethod=iterative cannot serve as a fixture: the reference's updateInSlice_iterative aborts with heap corruption --
    #  "malloc(): invalid size" -- in the first sweep of this very parameter set, IEEE build of oracle/ref_build)
    "o3_L4_woodbury": dict(args=dict(opdim=3, L=4, beta=2, s=10, delaySteps=6, sweeps=2, updateMethod="woodbury", sliceTrace=0),
                           keep=("meta", "sweep1_phi", "sweep2_phi", "sweep2_g", "sweep2_phiDelta", "rng_next")),
    # BASELINE config 1: the half-filled Hubbard model (src/dethubbard.cpp), plus checkerboard propagator, s not dividing m,
    # a larger lattice away from half filling
    "hub_L4": dict(harness="hubbard", args=dict(L=4, d=2, beta=2, dtau=0.1, s=10, t=1, U=4, mu=0, checkerboard=0, sweeps=4, measureSweeps=2)),
    "hub_L4_cb": dict(harness="hubbard", args=dict(L=4, d=2, beta=2, dtau=0.1, s=10, t=1, U=4, mu=0, checkerboard=1, sweeps=3, measureSweeps=1)),
    "hub_L4_s7": dict(harness="hubbard", args=dict(L=4, d=2, beta=2.3, dtau=0.1, s=7, t=1, U=2.5, mu=-0.2, checkerboard=0, sweeps=3, measureSweeps=1,
                                                   rngSeed=777, simindex=2)),
    "hub_L6": dict(harness="hubbard", args=dict(L=6, d=2, beta=3, dtau=0.1, s=10, t=1, U=6, mu=0.3, checkerboard=0, sweeps=2, measureSweeps=1)),
    # BASELINE config 3 (headline): checksums / subsamples only
    "o2_L16_b10": dict(args=dict(opdim=2, L=16, beta=10, s=10, delaySteps=16, sweeps=2, sliceTrace=0),
                       drop=("bchain_", "bdense", "bmult_", "init_coshTermPhi", "init_sinhTermPhi"),
                       subsample=16),
}


def run_case(name, spec):
    opdim = spec["args"].get("opdim", 2)
    exe = os.path.join(REFDIR, f"ref_harness_o{opdim}")
    if spec.get("harness") == "hubbard":
        exe = os.path.join(REFDIR, "ref_harness_hubbard")
    if not os.path.exists(exe):
        raise SystemExit(f"{exe} missing: run `make -C oracle/ref_build OPDIM={opdim}` (or `hubbard`) first")
    with tempfile.TemporaryDirectory() as td:
        cmd = [exe, td] + [f"{k}={v}" for k, v in spec["args"].items()]
        env = dict(os.environ, MKL_NUM_THREADS=str(spec.get("threads", 1)))
        subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL)
        arrays = {}
        for line in open(os.path.join(td, "manifest.txt")):
            parts = line.split()
            nm, dt, shape = parts[0], parts[1], tuple(int(x) for x in parts[2:])
            if any(nm.startswith(d) for d in spec.get("drop", ())):
                continue
            if spec.get("keep") and nm not in spec["keep"]:
                continue
            raw = np.fromfile(os.path.join(td, nm + ".bin"), dtype=np.complex128 if dt == "c16" else np.float64)
            a = raw.reshape(shape, order="F")          # harness writes column-major
            ss = spec.get("subsample")
            if ss and a.ndim == 2 and dt == "c16" and a.shape[0] > 64:
                arrays[nm + "_diag"] = np.diag(a).copy()
                arrays[nm + "_fro"] = np.array([np.linalg.norm(a)])
                a = a[::ss, ::ss].copy()
                nm = nm + f"_sub{ss}"
            if spec.get("hash_fields") and dt != "c16" and a.ndim == 3 and nm.endswith("_phi"):
                # long trajectories at the headline size: keep a SHA-256 of the field (raw bytes of the (N, OPDIM, m+1)
                # column-major cube) instead of 400 KB per sweep
                import hashlib
                arrays[nm + "_sha256"] = np.frombuffer(hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).digest(), dtype=np.uint8).copy()
                continue
            arrays[nm] = np.ascontiguousarray(a)
        stream = os.path.join(td, "configs-phi.binarystream")
        if os.path.exists(stream):
            arrays["cfgstream_phi_bytes"] = np.fromfile(stream, dtype=np.uint8)     # the file as the reference wrote it
        arrays["params_json"] = np.array(json.dumps(spec["args"]))
    os.makedirs(OUTDIR, exist_ok=True)
    out = os.path.join(OUTDIR, name + ".npz")
    np.savez_compressed(out, **arrays)
    print(f"{name}: {len(arrays)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


def make_rng():
    exe = os.path.join(REFDIR, "ref_harness_o2")
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([exe, td, "mode=rng"], check=True, stdout=subprocess.DEVNULL)
        arrays = {}
        for line in open(os.path.join(td, "manifest.txt")):
            nm = line.split()[0]
            arrays[nm] = np.fromfile(os.path.join(td, nm + ".bin"))
    np.savez_compressed(os.path.join(OUTDIR, "rng.npz"), **arrays)
    print("rng: ", list(arrays))


if __name__ == "__main__":
    which = sys.argv[1:] or (["rng"] + list(CASES))
    for nm in which:
        if nm == "rng":
            make_rng()
        else:
            run_case(nm, CASES[nm])
