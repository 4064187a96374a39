"""Replica exchange (parallel tempering) across GPUs: the build's DetQMCPT::replicaExchangeStep.

Reference: src/detqmcpt.h:963-1118 (gather control data + action at rank 0, serial sweep over adjacent
control parameters with rank 0's RNG, scatter of the new parameter index and the control data) and
src/detqmcpt.h:1122-1154 (consistency check).  The reference holds one replica per MPI rank; here a rank (= one
GPU) may hold several (the chains of a DetSDWBatch, all swept by one launch sequence), so "process p" of the
reference becomes the global replica number  rank * n_local + b.  Field configurations never move, only the
control parameter r and the small "control data" blob (MC step size adaptation + update statistics,
src/detsdwopdim.cpp:5219-5247) travel.

MI355X mapping: the reference's rooted gather / scatter pairs become ONE fixed-size all_gather
(8-byte action + control blob per rank) and ONE broadcast from rank 0 (new parameter index of every
rank + permuted control blobs) over torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  The payload is a few KB, i.e. latency bound; there is no data-path collective
inside a sweep.

The functions take any replica object with the reference's exchange surface
(get/set_exchange_parameter_value, get_exchange_action_contribution, get/set_control_data as bytes,
rand01), so the logic is testable on CPU with a stand-in replica.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np


def exchange_probability(par1, action1, par2, action2):
    """get_replica_exchange_probability<DetSDW> (src/detsdwopdim.cpp:5251-5264)."""
    delta = (par1 - par2) * (action2 - action1)
    return 1.0 if delta <= 0.0 else float(np.exp(-delta))


@dataclass
class ExchangeState:
    """Rank-0 bookkeeping of DetQMCPT (src/detqmcpt.h:151-181, :316-330) + the local parameter index."""
    controlParameterValues: list
    local_current_parameter_index: int
    current_process_par: list = field(default_factory=list)   # rank 0: process -> parameter index
    current_par_process: list = field(default_factory=list)   # rank 0: parameter index -> process
    par_swapUpProposed: list = field(default_factory=list)
    par_swapUpAccepted: list = field(default_factory=list)

    n_local: int = 1
    local_parameter_indices: list = field(default_factory=list)   # one per local replica

    @staticmethod
    def create(controlParameterValues, rank, world, n_local=1):
        nproc = world * n_local
        if len(controlParameterValues) != nproc:
            # src/detqmcpt.h:285-289
            raise ValueError("Number of processes %d does not match number of control parameter values %d"
                             % (nproc, len(controlParameterValues)))
        st = ExchangeState(list(controlParameterValues), rank * n_local)
        st.n_local = n_local
        st.local_parameter_indices = [rank * n_local + b for b in range(n_local)]
        if rank == 0:
            st.current_process_par = list(range(nproc))
            st.current_par_process = list(range(nproc))
            st.par_swapUpProposed = [0] * nproc
            st.par_swapUpAccepted = [0] * nproc
        return st


def control_data_to_bytes(cd):
    return bytes(memoryview(cd).cast("B")) if not isinstance(cd, (bytes, bytearray)) else bytes(cd)


class ReplicaAdapter:
    """Gives detqmc_amd.DetSDW the bytes-based control-data surface used here."""

    def __init__(self, rep):
        self.rep = rep

    def get_exchange_parameter_value(self):
        return self.rep.get_exchange_parameter_value()

    def set_exchange_parameter_value(self, v):
        self.rep.set_exchange_parameter_value(v)

    def get_exchange_action_contribution(self):
        return self.rep.get_exchange_action_contribution()

    def get_control_data(self):
        return bytes(bytearray(self.rep.get_control_data()))

    def set_control_data(self, blob):
        from ._lib import detsdw_control_data
        cd = detsdw_control_data.from_buffer_copy(blob)
        self.rep.set_control_data(cd)

    def rand01(self):
        return self.rep.rand01()


def replica_exchange_step(replica, state: ExchangeState, dist, device="cpu"):
    """One replicaExchangeStep (src/detqmcpt.h:963-1118).  Collective: call on every rank.  `replica` is one
    replica or the list of this rank's replicas (state.n_local of them); dist = None runs a single process
    (all replicas local, e.g. one GPU holding the whole ensemble).  Returns the new parameter index (list for a list)."""
    import torch
    single = not isinstance(replica, (list, tuple))
    reps = [replica] if single else list(replica)
    nl = len(reps)
    if nl != state.n_local:
        raise ValueError("expected %d local replicas, got %d" % (state.n_local, nl))
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    nproc = world * nl
    blobs_local = [r.get_control_data() for r in reps]
    nblob = len(blobs_local[0])
    rec = 8 + nblob
    # ---- all ranks -> everyone: per local replica [action as 8 bytes | control blob]
    payload = b"".join(np.float64(r.get_exchange_action_contribution()).tobytes() + bl for r, bl in zip(reps, blobs_local))
    send = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy()).to(device)
    if dist is not None:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
    else:
        recv = [send]
    # ---- rank 0 decides, serially over adjacent control parameters, with ITS first replica's RNG stream
    out = torch.zeros(nproc * rec, dtype=torch.uint8)
    if rank == 0:
        gathered = b"".join(r.cpu().numpy().tobytes() for r in recv)
        actions = [float(np.frombuffer(gathered[p * rec:p * rec + 8], dtype=np.float64)[0]) for p in range(nproc)]
        blobs = [gathered[p * rec + 8:(p + 1) * rec] for p in range(nproc)]
        for cpi1 in range(nproc - 1):
            cpi2 = cpi1 + 1
            par1, par2 = state.controlParameterValues[cpi1], state.controlParameterValues[cpi2]
            p1, p2 = state.current_par_process[cpi1], state.current_par_process[cpi2]
            prob = exchange_probability(par1, actions[p1], par2, actions[p2])
            state.par_swapUpProposed[cpi1] += 1
            if prob >= 1 or reps[0].rand01() <= prob:                      # :1041 (note: <=)
                state.par_swapUpAccepted[cpi1] += 1
                state.current_process_par[p1], state.current_process_par[p2] = cpi2, cpi1
                state.current_par_process[cpi1], state.current_par_process[cpi2] = p2, p1
                blobs[p1], blobs[p2] = blobs[p2], blobs[p1]
        buf = b"".join(np.int64(state.current_process_par[p]).tobytes() + blobs[p] for p in range(nproc))
        out = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).copy())
    out = out.to(device)
    if dist is not None:
        dist.broadcast(out, src=0)
    allb = out.cpu().numpy().tobytes()
    new_indices = []
    for b, rep in enumerate(reps):
        p = rank * nl + b
        mine = allb[p * rec:(p + 1) * rec]
        new_index = int(np.frombuffer(mine[:8], dtype=np.int64)[0])
        rep.set_exchange_parameter_value(state.controlParameterValues[new_index])
        rep.set_control_data(mine[8:])
        new_indices.append(new_index)
    state.local_parameter_indices = new_indices
    state.local_current_parameter_index = new_indices[0]
    return new_indices[0] if single else new_indices


def replica_exchange_consistency_check(replica, state: ExchangeState, dist, device="cpu"):
    """replicaExchangeConsistencyCheck (src/detqmcpt.h:1122-1154): every replica's r equals the table at
    rank 0 to 1e-10.  Debug aid -- the reference runs it every sweep, here it is opt-in."""
    import torch
    reps = list(replica) if isinstance(replica, (list, tuple)) else [replica]
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    send = torch.tensor([r.get_exchange_parameter_value() for r in reps], dtype=torch.float64, device=device)
    if dist is not None:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
    else:
        recv = [send]
    ok = True
    if rank == 0:
        vals = [float(v) for r in recv for v in r.cpu().tolist()]
        for p, v in enumerate(vals):
            want = state.controlParameterValues[state.current_process_par[p]]
            if abs(v - want) > 1e-10:
                ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    if dist is not None:
        dist.broadcast(flag, src=0)
    if int(flag.item()) != 1:
        raise RuntimeError("replica exchange consistency check failed")
    return True
