"""Worker for tests/test_pt_gloo.py: one rank of a 2-rank replica-exchange run on CPU (gloo).
The replica is a stand-in built on the CPU oracle (tests may use the oracle); the exchange logic under
test is detqmc_amd/pt.py, exactly what the GPU ranks run with backend nccl."""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


class OracleReplica:
    """Exchange surface of the reference's DetSDW (src/detsdwopdim.h:116-153) on top of the oracle."""

    def __init__(self, rank, rvalue, seed=4242, simindex=0):
        from detsdw_oracle import DetSDWOracle, SDWParams
        from dsfmt_oracle import RngWrapper
        rng = RngWrapper(seed, (simindex + 1) * (rank + 1))          # src/detqmcpt.h:301
        self.o = DetSDWOracle(SDWParams(opdim=2, L=4, beta=1.0, s=5, delaySteps=6, r=rvalue), rng=rng)

    def sweepThermalization(self):
        self.o.sweepThermalization()

    def get_exchange_parameter_value(self):
        return self.o.pars.r

    def set_exchange_parameter_value(self, v):
        self.o.pars.r = v

    def get_exchange_action_contribution(self):
        return self.o.get_exchange_action_contribution()

    def get_control_data(self):
        return struct.pack("<iidd", self.o.acceptedGlobalShifts, self.o.attemptedGlobalShifts, self.o.phiDelta,
                           self.o.lastAccRatioLocal_phi)

    def set_control_data(self, blob):
        a, b, pd, la = struct.unpack("<iidd", blob)
        self.o.acceptedGlobalShifts, self.o.attemptedGlobalShifts, self.o.phiDelta = a, b, pd
        self.o.lastAccRatioLocal_phi = la

    def rand01(self):
        return self.o.rng.rand01()


def main():
    import torch.distributed as dist
    from detqmc_amd.pt import (ExchangeState, ObservableRouterPT, replica_exchange_step, replica_exchange_consistency_check,
                               write_exchange_statistics)
    out = sys.argv[1]
    rvalues = json.loads(sys.argv[2])
    steps = int(sys.argv[3])
    n_local = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    procs = [rank * n_local + b for b in range(n_local)]          # global replica numbers held by this rank
    reps = [OracleReplica(p, rvalues[p]) for p in procs]
    for p, rep in zip(procs, reps):
        rep.o.phiDelta = 0.5 + 0.1 * p              # make the control data distinguishable
    st = ExchangeState.create(rvalues, rank, world, n_local)
    hist = [[] for _ in reps]
    # observable routing: every replica "measures" after each sweep; rank 0 files the values under the
    # control parameter the replica holds at that moment
    router = ObservableRouterPT(st, ["rHeld", "action"], [("vec", 3)], sweeps=steps, jk_blocks=steps if steps % 3 else 3, timeseries=True)
    for it in range(steps):
        for rep in reps:
            rep.sweepThermalization()
        vals = []
        for rep in reps:
            r_, a_ = rep.get_exchange_parameter_value(), rep.get_exchange_action_contribution()
            vals.append(({"rHeld": r_, "action": a_}, {"vec": [r_, r_ * r_, a_]}))
        router.insert(it, vals, dist)
        idx = replica_exchange_step(reps if n_local > 1 else reps[0], st, dist)
        replica_exchange_consistency_check(reps, st, dist)
        idx = idx if n_local > 1 else [idx]
        for b, rep in enumerate(reps):
            hist[b].append(dict(index=idx[b], r=rep.get_exchange_parameter_value(), phiDelta=rep.o.phiDelta,
                                action=rep.get_exchange_action_contribution()))
    res = dict(rank=rank, hist=hist[0], hist_all=hist)
    if rank == 0:
        res["proposed"] = st.par_swapUpProposed
        res["accepted"] = st.par_swapUpAccepted
        res["process_par"] = st.current_process_par
        res["going_up"], res["going_down"] = st.par_countGoingUp, st.par_countGoingDown
        res["routed"] = [[list(x) for x in router.evaluate_jackknife(c)] for c in range(len(rvalues))]
        router.write_results(out, "r", {"L": 4, "opdim": 2, "r": "overwritten"}, {"sweeps": steps}, {"controlParameterName": "r"})
        write_exchange_statistics(st, out, [{"L": 4}, {"sweeps": steps}])
    json.dump(res, open(os.path.join(out, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
