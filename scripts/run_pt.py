#!/usr/bin/env python3
"""Replica-exchange (parallel tempering in r) driver: the build's maindetqmcptsdwopdim on N GPUs.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        scripts/run_pt.py --rmin -1.4 --rmax -0.6 --per-gpu 8 --sweeps 20 --exchange-interval 1
    python scripts/run_pt.py ...                      # single process: the whole ensemble on one GPU

Every rank (= GPU) holds `--per-gpu` replicas as ONE DetSDWBatch (all swept in lockstep, one launch for all);
global replica p = rank * per_gpu + b starts at control parameter r_p of a linear ladder, like the reference's
controlParameterValues (src/detqmcpt.h:285-330).  After every `--exchange-interval` sweeps the ranks run
replicaExchangeStep (detqmc_amd/pt.py: one all_gather + one broadcast of a few KB over RCCL).
After `--thermalization` sweeps, `--sweeps` measurement sweeps follow (sweep(true) every `--measure-interval`): every replica's
observables are filed under the control parameter it holds at that moment (ObservableRouterPT) and rank 0 writes the reference's
output tree into `--out`: p<cpi>_r<value>/results.values, results-<vector>.values, exchange-{parameters,acceptance,diffusion}.values
(src/mpiobservablehandlerpt.cpp:221-300, src/detqmcpt.h:596-660).
`--backend gloo --one-device` rehearses several ranks on a one-GPU box (collectives over gloo with host tensors).
Prints the parameter index every replica ends at and the swap acceptance per neighbouring pair.
"""
import argparse
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rmin", type=float, default=-1.4)
    ap.add_argument("--rmax", type=float, default=-0.6)
    ap.add_argument("--per-gpu", type=int, default=8)
    ap.add_argument("--sweeps", type=int, default=10)
    ap.add_argument("--exchange-interval", type=int, default=1)
    ap.add_argument("--L", type=int, default=16)
    ap.add_argument("--beta", type=float, default=10.0)
    ap.add_argument("--opdim", type=int, default=2)
    ap.add_argument("--stabilisation", default="qr")
    ap.add_argument("--check", action="store_true", help="run the consistency check after every exchange")
    ap.add_argument("--thermalization", type=int, default=0, help="thermalization sweeps before the --sweeps measurement sweeps")
    ap.add_argument("--measure-interval", type=int, default=1)
    ap.add_argument("--jk-blocks", type=int, default=1)
    ap.add_argument("--out", default=None, help="directory for the per-parameter results tree (rank 0)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true", help="all ranks drive GPU 0 (rehearsal on a one-GPU box, needs --backend gloo)")
    ap.add_argument("--conf", default=None, help="the reference's simulation.conf (mpimaindetqmcptsdwopdim): run THAT simulation -- thermalization, "
                    "measurement sweeps, replica exchange, per-parameter results / time series / configuration streams -- and write the "
                    "reference's output tree into --out (default: the directory of the file)")
    a = ap.parse_args()
    if a.conf:
        return run_conf(a)

    import torch
    from detqmc_amd import DetSDWBatch, SDWParams
    from detqmc_amd.pt import (ExchangeState, ObservableRouterPT, ReplicaAdapter, replica_exchange_step,
                               replica_exchange_consistency_check, write_exchange_statistics)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    device = "cpu"
    if a.one_device:
        local = 0
    if world > 1 or "RANK" in os.environ:         # under torch.distributed.run the collectives run for ONE rank as well
        import torch.distributed as dist
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            device = "cuda"
        else:
            dist.init_process_group("gloo")
    nproc = world * a.per_gpu
    rvals = [a.rmin + (a.rmax - a.rmin) * p / max(nproc - 1, 1) for p in range(nproc)]
    p0 = SDWParams(opdim=a.opdim, L=a.L, beta=a.beta, s=10, delaySteps=16, device=local, stabilisation=a.stabilisation,
                   fermionMeasurements=True)
    mine = [rank * a.per_gpu + b for b in range(a.per_gpu)]
    batch = DetSDWBatch([dataclasses.replace(p0, r=rvals[p], simindex=p) for p in mine])
    reps = [ReplicaAdapter(batch.chain(b)) for b in range(a.per_gpu)]
    st = ExchangeState.create(rvals, rank, world, a.per_gpu)
    def exchange(sw):
        if sw % a.exchange_interval == 0:
            replica_exchange_step(reps, st, dist, device=device)
            if a.check:
                replica_exchange_consistency_check(reps, st, dist, device=device)

    for sw in range(1, a.thermalization + 1):
        batch.sweepThermalization()
        exchange(sw)
    N = a.L * a.L
    scalars = ["normMeanPhi", "associatedEnergy"] + (["phiRhoS_Gs", "phiRhoS_Gc"] if a.opdim == 2 else []) + \
              ["pairPlusMax", "pairMinusMax", "greenK0", "greenLocal", "occDiffSq"]
    vectors = [("kOccX", N), ("kOccY", N), ("pairPlus", N), ("pairMinus", N)]
    router = ObservableRouterPT(st, scalars, vectors, sweeps=a.sweeps, jk_blocks=a.jk_blocks, measure_interval=a.measure_interval)
    for sw in range(1, a.sweeps + 1):
        measure = a.out is not None and sw % a.measure_interval == 0
        if a.out is None and a.thermalization == 0:
            batch.sweepThermalization()               # the round-1 behaviour: thermalization sweeps only
        else:
            batch.sweep(measure)
        if measure:
            vals = []
            for b in range(a.per_gpu):
                c = batch.chain(b)
                o = c.observables
                vals.append(({n: getattr(o, n) for n in scalars}, {n: c.observable_vector(n) for n, _ in vectors}))
            router.insert(sw - 1, vals, dist, device)
        exchange(a.thermalization + sw)
    if a.out is not None and rank == 0:
        os.makedirs(a.out, exist_ok=True)
        meta_model = {"model": "sdw", "opdim": a.opdim, "L": a.L, "beta": a.beta, "s": 10, "r": "-"}
        meta_mc = {"sweeps": a.sweeps, "thermalization": a.thermalization, "measureInterval": a.measure_interval, "jkBlocks": a.jk_blocks}
        meta_pt = {"controlParameterName": "r", "exchangeInterval": a.exchange_interval}
        router.write_results(a.out, "r", meta_model, meta_mc, meta_pt)
        write_exchange_statistics(st, a.out, [meta_model, meta_mc, meta_pt])
    print("rank %d: replicas %s hold parameter indices %s" % (rank, mine, st.local_parameter_indices), flush=True)
    if rank == 0:
        acc = [("%d/%d" % (x, y)) for x, y in zip(st.par_swapUpAccepted[:-1], st.par_swapUpProposed[:-1])]
        print("swap up accepted/proposed per parameter pair: " + " ".join(acc), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_conf(a):
    """DetQMCPT<DetSDW>::run (src/detqmcpt.h:761-958) for the reference's own configuration file: every rank holds
    len(rValues) / world replicas in one DetSDWBatch; global replica p has the RNG stream of the reference's process p
    (RngWrapper(rngSeed, (simindex + 1) (p + 1)), src/detqmcpt.h:301)."""
    import torch
    from detqmc_amd import DetSDWBatch, SDWParams
    from detqmc_amd import pt as PT

    conf = PT.parse_simulation_conf(a.conf)
    out = a.out or os.path.dirname(os.path.abspath(a.conf))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist, device = None, "cpu"
    if world > 1 or "RANK" in os.environ:         # under torch.distributed.run the collectives run for ONE rank as well
        import torch.distributed as dist
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            device = "cuda"
        else:
            dist.init_process_group("gloo")
    rvals = [float(v) for v in (conf["rValues"] if isinstance(conf["rValues"], list) else [conf["rValues"]])]
    nproc = len(rvals)
    if nproc % world:
        raise SystemExit("Number of processes %d does not divide the number of control parameter values %d" % (world, nproc))
    per = nproc // world
    g = lambda k, d: conf.get(k, d)
    if g("model", "sdw") != "sdw" or g("spinProposalMethod", "box") != "box" or PT._b(g("turnoffFermions", "false")):
        raise SystemExit("run_pt.py --conf: only model = sdw with box proposals and fermions switched on")
    simindex = int(g("simindex", 0))
    kw = dict(opdim=int(g("opdim", 3)), L=int(g("L", 4)), dtau=float(g("dtau", 0.1)), s=int(g("s", 1)), c=float(g("c", 1.0)),
              u=float(g("u", 1.0)), lambda_=float(g("lambda", 1.0)), txhor=float(g("txhor", -1.0)), txver=float(g("txver", -0.5)),
              tyhor=float(g("tyhor", 0.5)), tyver=float(g("tyver", 1.0)), mu=float(g("mu", 0.5)), accRatio=float(g("accRatio", 0.5)),
              delaySteps=int(g("delaySteps", 16)), updateMethod=g("updateMethod", "iterative"), bc=g("bc", "pbc"),
              weakZflux=PT._b(g("weakZflux", "false")), globalShift=PT._b(g("globalShift", "false")),
              wolffClusterUpdate=PT._b(g("wolffClusterUpdate", "false")), wolffClusterShiftUpdate=PT._b(g("wolffClusterShiftUpdate", "false")),
              globalUpdateInterval=int(g("globalUpdateInterval", 100)), checkerboard=PT._b(g("checkerboard", "false")),
              fermionMeasurements=not PT._b(g("turnoffFermionMeasurements", "false")), rngSeed=int(g("rngSeed", 0)),
              cdwU=float(g("cdwU", 0.0)), stabilisation=a.stabilisation, device=local)
    if "m" in conf:
        kw["m"] = int(conf["m"])
    else:
        kw["beta"] = float(conf["beta"])
    p0 = SDWParams(**kw)
    mine = [rank * per + b for b in range(per)]
    batch = DetSDWBatch([dataclasses.replace(p0, r=rvals[p], simindex=(simindex + 1) * (p + 1) - 1) for p in mine])
    reps = [PT.ReplicaAdapter(batch.chain(b)) for b in range(per)]
    st = PT.ExchangeState.create(rvals, rank, world, per)
    therm, sweeps = int(g("thermalization", 0)), int(g("sweeps", 0))
    mi, xi = int(g("measureInterval", 1)), int(g("exchangeInterval", 1))
    csi = int(g("saveConfigurationStreamInterval", mi))
    save_bin = PT._b(g("saveConfigurationStreamBinary", "false"))
    timeseries = PT._b(g("timeseries", "false"))
    opdim, N = kw["opdim"], kw["L"] ** 2
    scalars = ["normMeanPhi", "associatedEnergy"] + (["phiRhoS_Gs", "phiRhoS_Gc"] if opdim == 2 else [])
    vectors = []
    if kw["fermionMeasurements"]:
        scalars += ["pairPlusMax", "pairMinusMax", "greenK0", "greenLocal", "occDiffSq"]
        vectors = [("kOccX", N), ("kOccY", N), ("pairPlus", N), ("pairMinus", N)]
    router = PT.ObservableRouterPT(st, scalars, vectors, sweeps=sweeps, jk_blocks=int(g("jkBlocks", 1)), measure_interval=mi, timeseries=timeseries)
    meta_model, meta_mc, meta_pt = PT.reference_metadata(conf, rvals)
    subdir = lambda cpi: os.path.join(out, PT.control_parameter_subdir(cpi, "r", rvals[cpi]))

    def exchange(done):
        if xi != 0 and done % xi == 0:
            PT.replica_exchange_step(reps, st, dist, device=device)
        if a.check:
            PT.replica_exchange_consistency_check(reps, st, dist, device=device)

    for sw in range(1, therm + 1):                                       # stage T
        batch.sweepThermalization()
        exchange(sw)
    if save_bin:                                                          # setup_SaveConfigurations (:662-699)
        for cpi in st.local_parameter_indices if world > 1 else range(nproc):
            os.makedirs(subdir(cpi), exist_ok=True)
    sw_counter = 0
    for done in range(sweeps):                                            # stage M; `done` = sweepsDone before the sweep
        sw_counter += 1
        measure = sw_counter % mi == 0
        batch.sweep(measure)
        if measure:
            vals = []
            for b in range(per):
                c = batch.chain(b)
                o = c.observables
                vals.append(({n: getattr(o, n) for n in scalars}, {n: c.observable_vector(n) for n, _ in vectors}))
            router.insert(done, vals, dist, device)
            if save_bin and sw_counter % csi == 0:
                # buffer_local_system_configuration / gather_and_output_buffered_system_configurations (:703-757): the
                # configuration is filed under the control parameter its replica holds NOW
                for b in range(per):
                    d = subdir(st.local_parameter_indices[b])
                    os.makedirs(d, exist_ok=True)
                    batch.chain(b).saveConfigurationStreamBinary(d)
        if done + 1 < sweeps:                                             # the last sweep ends the loop: stage F, no exchange (:946-956)
            exchange(therm + done + 1)
    if rank == 0:
        os.makedirs(out, exist_ok=True)
        router.write_results(out, "r", meta_model, meta_mc, meta_pt)
        PT.write_timeseries(router, out, "r", meta_model, meta_mc, meta_pt)
        PT.write_exchange_statistics(st, out, [{k: v for k, v in meta_model.items() if k != "r"}, meta_mc, meta_pt])
        if save_bin:
            for cpi in range(nproc):
                mm = dict(meta_model)
                mm["r"] = PT.num_to_string(rvals[cpi])
                PT.write_config_infoheader(subdir(cpi), mm, meta_mc, meta_pt, cdw=kw["cdwU"] != 0.0)
        print("Measurements finished", flush=True)
        print("exchange payload: %s tensors, backend %s" % (getattr(st, "exchange_payload", "none"), a.backend if dist is not None else "none (single process)"), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
