"""Worker for test_replica_exchange_across_two_processes_with_real_chains (tests/test_gpu_parity.py): one rank of a 2-rank
replica-exchange run whose replicas are REAL batched chains on the GPU (both ranks drive device 0 on the one-GPU test box;
on a multi-GPU node rank = device).  Collectives over gloo with CPU tensors -- the exchange payload is a few hundred bytes;
`--backend nccl` runs the same code over RCCL."""
import dataclasses
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch.distributed as dist
    from detqmc_amd import DetSDWBatch, SDWParams
    from detqmc_amd.pt import (ExchangeState, ObservableRouterPT, ReplicaAdapter, replica_exchange_step,
                               replica_exchange_consistency_check)
    out, rvalues, steps, n_local = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = "cpu"
    if world > 1:
        if backend == "nccl":
            import torch
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            device = "cuda"
        dist.init_process_group(backend)
    d = dist if world > 1 else None
    gpu = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("DQMC_PT_ONE_DEVICE_PER_RANK") else 0
    p0 = SDWParams(opdim=2, L=4, beta=2.0, s=10, delaySteps=6, stabilisation="qr", globalShift=True, globalUpdateInterval=2, device=gpu)
    procs = [rank * n_local + b for b in range(n_local)]
    batch = DetSDWBatch([dataclasses.replace(p0, r=rvalues[p], simindex=p) for p in procs])
    reps = [ReplicaAdapter(batch.chain(b)) for b in range(n_local)]
    st = ExchangeState.create(rvalues, rank, world, n_local)
    router = ObservableRouterPT(st, ["normMeanPhi", "associatedEnergy"], [], sweeps=steps, jk_blocks=2)
    hist = [[] for _ in procs]
    for it in range(steps):
        batch.sweep(True)
        vals = []
        for b in range(n_local):
            o = batch.chain(b).observables
            vals.append(({"normMeanPhi": o.normMeanPhi, "associatedEnergy": o.associatedEnergy}, {}))
        router.insert(it, vals, d, device)
        idx = replica_exchange_step(reps, st, d, device)
        replica_exchange_consistency_check(reps, st, d, device)
        for b in range(n_local):
            c = batch.chain(b)
            hist[b].append(dict(index=idx[b], r=c.get_exchange_parameter_value(), phiDelta=c.info.phiDelta,
                                phi=hashlib.sha256(np.ascontiguousarray(c.phi).tobytes()).hexdigest()))
    res = dict(rank=rank, hist=hist)
    if rank == 0:
        res.update(proposed=st.par_swapUpProposed, accepted=st.par_swapUpAccepted,
                   routed=[[list(x) for x in router.evaluate_jackknife(c)] for c in range(len(rvalues))])
    json.dump(res, open(os.path.join(out, "rank%d_of%d.json" % (rank, world)), "w"))
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
