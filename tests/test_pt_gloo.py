"""Multi-process (world_size 2, gloo, CPU) test of the replica-exchange step: detqmc_amd/pt.py against a
serial restatement of the reference's replicaExchangeStep (src/detqmcpt.h:963-1118)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _serial_expectation(rvalues, steps):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pt_worker import OracleReplica
    from detsdw_oracle import replica_exchange_probability
    world = len(rvalues)
    reps = [OracleReplica(p, rvalues[p]) for p in range(world)]
    for p, r in enumerate(reps):
        r.o.phiDelta = 0.5 + 0.1 * p
    process_par = list(range(world))
    par_process = list(range(world))
    hist = [[] for _ in range(world)]
    for it in range(steps):
        for r in reps:
            r.sweepThermalization()
        actions = [r.get_exchange_action_contribution() for r in reps]
        blobs = [r.get_control_data() for r in reps]
        for c1 in range(world - 1):
            c2 = c1 + 1
            p1, p2 = par_process[c1], par_process[c2]
            prob = replica_exchange_probability(rvalues[c1], actions[p1], rvalues[c2], actions[p2])
            if prob >= 1 or reps[0].rand01() <= prob:
                process_par[p1], process_par[p2] = c2, c1
                par_process[c1], par_process[c2] = p2, p1
                blobs[p1], blobs[p2] = blobs[p2], blobs[p1]
        for p, r in enumerate(reps):
            r.set_exchange_parameter_value(rvalues[process_par[p]])
            r.set_control_data(blobs[p])
            hist[p].append(dict(index=process_par[p], r=r.get_exchange_parameter_value(), phiDelta=r.o.phiDelta))
    return hist


@pytest.mark.parametrize("rvalues", [[-1.0, -0.9], [0.5, -2.5]])
def test_replica_exchange_two_ranks_gloo(tmp_path, rvalues):
    steps = 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() % 500)), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "pt_worker.py"), str(tmp_path),
           json.dumps(rvalues), str(steps)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    got = [json.load(open(tmp_path / ("rank%d.json" % p))) for p in range(2)]
    want = _serial_expectation(rvalues, steps)
    swapped = False
    for p in range(2):
        for it in range(steps):
            g, w = got[p]["hist"][it], want[p][it]
            assert g["index"] == w["index"] and g["r"] == w["r"] and g["phiDelta"] == w["phiDelta"], (p, it, g, w)
            swapped |= g["index"] != p
        # every parameter value is held by exactly one rank after each step
    for it in range(steps):
        assert sorted(got[p]["hist"][it]["index"] for p in range(2)) == [0, 1]
    assert got[0]["proposed"] == [steps, 0]
    assert got[0]["accepted"][0] >= (1 if swapped else 0)
    # control data travels with the PARAMETER, not with the rank (src/detqmcpt.h:1048-1050)
    for p in range(2):
        last = got[p]["hist"][-1]
        assert abs(last["phiDelta"] - (0.5 + 0.1 * last["index"])) < 0.11


def test_replica_exchange_two_ranks_two_local_replicas_gloo(tmp_path):
    """2 ranks x 2 replicas per rank (what a GPU holding a batch of chains does): global replica p = rank * 2 + b
    takes the place of the reference's process p."""
    rvalues, steps = [-1.0, -0.9, -0.8, -0.7], 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + ((os.getpid() + 7) % 500)), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "pt_worker.py"), str(tmp_path),
           json.dumps(rvalues), str(steps), "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    got = [json.load(open(tmp_path / ("rank%d.json" % p))) for p in range(2)]
    want = _serial_expectation(rvalues, steps)
    for rank in range(2):
        for b in range(2):
            p = rank * 2 + b
            for it in range(steps):
                g, w = got[rank]["hist_all"][b][it], want[p][it]
                assert g["index"] == w["index"] and g["r"] == w["r"] and g["phiDelta"] == w["phiDelta"], (p, it, g, w)
    for it in range(steps):
        assert sorted(got[r]["hist_all"][b][it]["index"] for r in range(2) for b in range(2)) == [0, 1, 2, 3]
    assert got[0]["proposed"] == [steps, steps, steps, 0]


def test_single_process_exchange_between_local_replicas():
    """dist = None: the whole ensemble on one rank (one GPU), no collective at all."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pt_worker import OracleReplica
    from detqmc_amd.pt import ExchangeState, replica_exchange_step, replica_exchange_consistency_check
    rvalues, steps = [0.5, -2.5], 2
    reps = [OracleReplica(p, rvalues[p]) for p in range(2)]
    for p, r in enumerate(reps):
        r.o.phiDelta = 0.5 + 0.1 * p
    st = ExchangeState.create(rvalues, 0, 1, n_local=2)
    want = _serial_expectation(rvalues, steps)
    for it in range(steps):
        for r in reps:
            r.sweepThermalization()
        idx = replica_exchange_step(reps, st, None)
        replica_exchange_consistency_check(reps, st, None)
        for p in range(2):
            assert idx[p] == want[p][it]["index"] and reps[p].get_exchange_parameter_value() == want[p][it]["r"]


def test_rank_count_must_match_parameter_count():
    from detqmc_amd.pt import ExchangeState
    with pytest.raises(ValueError):
        ExchangeState.create([-1.0, -0.5, 0.0], rank=0, world=2)


# ------------------------------------------------------------------------------------------------
# observable routing by control parameter + exchange statistics
# ------------------------------------------------------------------------------------------------
def _reference_jackknife(series, jk_blocks):
    """jackknifeBlockEstimates + jackknife as the reference computes them from a time series
    (src/statistics.h:51-75, 132-144); mean = plain average of all samples"""
    import numpy as np
    series = np.asarray(series, dtype=float)
    n = len(series)
    block = n // jk_blocks
    est = np.zeros((jk_blocks,) + series.shape[1:])
    for i in range(jk_blocks * block):
        for jb in range(jk_blocks):
            if jb != i // block:
                est[jb] += series[i]
    est /= (jk_blocks * block - block)
    mean = series.mean(axis=0)
    return mean, np.sqrt((jk_blocks - 1.0) / jk_blocks * np.sum((mean[None] - est) ** 2, axis=0))


def test_observables_are_routed_to_the_control_parameter_two_ranks_gloo(tmp_path):
    """2 ranks x 2 replicas: each replica reports (the r it holds, its action, a 3-vector) after every sweep; the values
    must land under the control parameter index the replica holds at that moment (ObservableHandlerPTCommon::handleValues,
    src/mpiobservablehandlerpt.h:171-186), with the reference's jackknife, and the output tree must be the reference's
    p<cpi>_r<value>/results.values + exchange-*.values."""
    import numpy as np
    rvalues, steps = [-1.0, -0.9, -0.8, -0.7], 6
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + ((os.getpid() + 13) % 500)), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "pt_worker.py"), str(tmp_path),
           json.dumps(rvalues), str(steps), "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    got = [json.load(open(tmp_path / ("rank%d.json" % p))) for p in range(2)]
    # serial replay: the parameter index a replica holds DURING sweep `it` is the one set by the exchange after sweep it-1
    hist = {rank * 2 + b: got[rank]["hist_all"][b] for rank in range(2) for b in range(2)}
    series = {c: [] for c in range(4)}
    moved = False
    for it in range(steps):
        for p in range(4):
            held = p if it == 0 else hist[p][it - 1]["index"]
            moved |= held != p
            r_, a_ = rvalues[held], hist[p][it]["action"]
            series[held].append([r_, a_, r_, r_ * r_, a_])
    assert moved, "no exchange was accepted: the routing would not be exercised"
    jk = 3
    for c in range(4):
        assert len(series[c]) == steps                     # every parameter is held by exactly one replica in every sweep
        mean, err = _reference_jackknife(series[c], jk)
        gm, ge = (np.array(x) for x in got[0]["routed"][c])
        assert np.allclose(gm, mean, rtol=1e-13, atol=1e-13) and np.allclose(ge, err, rtol=1e-9, atol=1e-13), c
        assert abs(gm[0] - rvalues[c]) < 1e-15 and ge[0] < 1e-13       # "the r it holds" filed under r: constant
        sub = tmp_path / ("p%d_r%s" % (c, ("%g" % rvalues[c])))
        lines = open(sub / "results.values").read().splitlines()
        assert lines[0] == "## Monte Carlo results for observable expectation values"
        assert "# r = %g" % rvalues[c] in lines and "# key = observable" in lines and lines[-3].startswith("## observable")
        rows = [l.split("\t") for l in lines if not l.startswith("#")]
        assert [r[0] for r in rows] == ["action", "rHeld"]
        assert abs(float(rows[0][1]) - mean[1]) <= 1e-14 * abs(mean[1]) and abs(float(rows[1][1]) - rvalues[c]) < 1e-15
        vrows = [l.split("\t") for l in open(sub / "results-vec.values").read().splitlines() if not l.startswith("#")]
        assert len(vrows) == 3 and abs(float(vrows[1][1]) - rvalues[c] ** 2) < 1e-14
    # exchange statistics files (src/detqmcpt.h:596-651)
    prop, acc = got[0]["proposed"], got[0]["accepted"]
    arows = [l.split("\t") for l in open(tmp_path / "exchange-acceptance.values").read().splitlines() if not l.startswith("#")]
    assert [int(r[0]) for r in arows] == [0, 1, 2, 3]
    for c in range(4):
        assert abs(float(arows[c][1]) - (acc[c] / prop[c] if prop[c] else 0.0)) < 1e-14
    prow = [l.split("\t") for l in open(tmp_path / "exchange-parameters.values").read().splitlines() if not l.startswith("#")]
    assert [float(r[1]) for r in prow] == rvalues
    # diffusion histogram replayed from the exchange history (src/detqmcpt.h:1016-1029)
    going = {p: 0 for p in range(4)}
    up, down = [0] * 4, [0] * 4
    for it in range(steps):
        for p in range(4):
            held = p if it == 0 else hist[p][it - 1]["index"]
            if held == 3:
                going[p] = -1
            elif held == 0:
                going[p] = +1
            if going[p] == -1:
                down[held] += 1
            elif going[p] == +1:
                up[held] += 1
    assert got[0]["going_up"] == up and got[0]["going_down"] == down
    drow = [l.split("\t") for l in open(tmp_path / "exchange-diffusion.values").read().splitlines() if not l.startswith("#")]
    for c in range(4):
        assert abs(float(drow[c][1]) - (up[c] / (up[c] + down[c]) if up[c] + down[c] else 0.0)) < 1e-14


def test_reference_metadata_headers_match_the_reference_output_tree(tmp_path):
    """parse_simulation_conf + reference_metadata + the data-map writers reproduce, line by line, the headers of the files the
    reference's detqmcptsdwo2 wrote for tests/golden/detqmcpt_run_o2_L4/simulation.conf (DataMapWriter,
    /root/reference/src/datamapwriter.h:116-160; metadata maps src/detsdwopdim.cpp:363-438, src/detqmcparams.cpp:60-98,
    src/detqmcptparams.cpp:40-68); the numbers are checked on the GPU (tests/test_gpu_detqmc_shim.py)."""
    import os
    import numpy as np
    from detqmc_amd import pt as PT
    case = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detqmcpt_run_o2_L4")
    conf = PT.parse_simulation_conf(os.path.join(case, "simulation.conf"))
    rvals = [float(v) for v in conf["rValues"]]
    assert rvals == [-1.3, -1.2, -1.1, -1.0]
    mm, mc, mp = PT.reference_metadata(conf, rvals)
    st = PT.ExchangeState.create(rvals, 0, 1, 4)
    scal = ["normMeanPhi", "associatedEnergy", "phiRhoS_Gs", "phiRhoS_Gc", "pairPlusMax", "pairMinusMax", "greenK0", "greenLocal", "occDiffSq"]
    router = PT.ObservableRouterPT(st, scal, [("kOccX", 16), ("kOccY", 16), ("pairPlus", 16), ("pairMinus", 16)], sweeps=20, jk_blocks=5,
                                   measure_interval=2, timeseries=True)
    rng = np.random.default_rng(1)
    for sw in range(1, 20, 2):
        router.insert(sw, [({n: rng.random() for n in scal}, {n: rng.random(16) for n in ("kOccX", "kOccY", "pairPlus", "pairMinus")})
                           for _ in range(4)], None)
    router.write_results(tmp_path, "r", mm, mc, mp)
    PT.write_timeseries(router, tmp_path, "r", mm, mc, mp)
    PT.write_exchange_statistics(st, tmp_path, [{k: v for k, v in mm.items() if k != "r"}, mc, mp])
    for cpi in range(4):
        m2 = dict(mm)
        m2["r"] = PT.num_to_string(rvals[cpi])
        PT.write_config_infoheader(os.path.join(tmp_path, PT.control_parameter_subdir(cpi, "r", rvals[cpi])), m2, mc, mp)
    hdr = lambda p: [l for l in open(p) if l.startswith("#")]
    nfiles = 0
    for dirpath, _, files in os.walk(os.path.join(case, "expected")):
        rel = os.path.relpath(dirpath, os.path.join(case, "expected"))
        for fn in files:
            if fn.endswith(".binarystream"):
                continue
            got = os.path.join(tmp_path, rel, fn)
            assert os.path.exists(got), os.path.join(rel, fn)
            assert hdr(got) == hdr(os.path.join(dirpath, fn)), os.path.join(rel, fn)
            nfiles += 1
    assert nfiles == 4 * 15 + 3
    # single-block error estimate: standard deviation with N - 1 (variance(), src/statistics.h:36-46)
    r1 = PT.ObservableRouterPT(st, ["x"], [], sweeps=4, jk_blocks=1, measure_interval=1, timeseries=True)
    xs = [[0.1, 0.4, 0.2, 0.9], [1.0, 2.0, 4.0, 8.0], [0, 0, 0, 1.0], [3.0, 3.0, 3.0, 3.0]]
    for sw in range(4):
        r1.insert(sw, [({"x": xs[p][sw]}, {}) for p in range(4)], None)
    for p in range(4):
        mean, err = r1.evaluate_jackknife(p)
        assert abs(mean[0] - np.mean(xs[p])) < 1e-15 and abs(err[0] - np.std(xs[p], ddof=1)) < 1e-15
