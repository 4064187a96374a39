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
    a = ap.parse_args()

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
    if world > 1:
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


if __name__ == "__main__":
    main()
