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
    # replica diffusion in parameter space (ExchangeStatistics, src/detqmcpt.h:92-115): 0 = NONE_P, +1 = UP_P, -1 = DOWN_P
    process_goingWhere: list = field(default_factory=list)
    par_countGoingUp: list = field(default_factory=list)
    par_countGoingDown: list = field(default_factory=list)

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
            st.process_goingWhere = [0] * nproc
            st.par_countGoingUp = [0] * nproc
            st.par_countGoingDown = [0] * nproc
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
    # ---- all ranks -> everyone: per local replica the exchange action (8 bytes) and the control blob.
    # Device collectives (backend nccl = RCCL): when the local replicas are the chains of ONE DetSDWBatch, their actions are reduced on
    # the GPU straight into the send tensor of the all_gather (detsdw_exchange_actions_device) -- no device -> host -> device hop;
    # the control blobs (host-side statistics + the step-size state) travel in a second all_gather.
    batch = getattr(getattr(reps[0], "rep", None), "_batch", None)
    on_device = (str(device) != "cpu" and batch is not None and len(batch) == nl
                 and all(getattr(getattr(r, "rep", None), "_batch", None) is batch and r.rep._chain == b for b, r in enumerate(reps)))
    state.exchange_payload = "device" if on_device else "host"       # which path the last step took (drivers report it)
    if on_device:
        send_act = torch.empty(nl, dtype=torch.float64, device=device)
        batch.exchange_actions_device(send_act.data_ptr())
        send_blob = torch.from_numpy(np.frombuffer(b"".join(blobs_local), dtype=np.uint8).copy()).to(device)
        if dist is not None:
            recv_act = [torch.empty_like(send_act) for _ in range(world)]
            recv_blob = [torch.empty_like(send_blob) for _ in range(world)]
            dist.all_gather(recv_act, send_act)
            dist.all_gather(recv_blob, send_blob)
        else:
            recv_act, recv_blob = [send_act], [send_blob]
    else:
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
        if on_device:
            actions = [float(v) for t in recv_act for v in t.cpu().tolist()]
            allblob = b"".join(t.cpu().numpy().tobytes() for t in recv_blob)
            blobs = [allblob[p * nblob:(p + 1) * nblob] for p in range(nproc)]
        else:
            gathered = b"".join(r.cpu().numpy().tobytes() for r in recv)
            actions = [float(np.frombuffer(gathered[p * rec:p * rec + 8], dtype=np.float64)[0]) for p in range(nproc)]
            blobs = [gathered[p * rec + 8:(p + 1) * rec] for p in range(nproc)]
        # histograms of replicas moving up or down in parameter space (src/detqmcpt.h:1016-1029): a replica that visited
        # the highest parameter last is "going down", the lowest "going up"
        for pi in range(nproc):
            npar = state.current_process_par[pi]
            if npar == nproc - 1:
                state.process_goingWhere[pi] = -1
            elif npar == 0:
                state.process_goingWhere[pi] = +1
            if state.process_goingWhere[pi] == -1:
                state.par_countGoingDown[npar] += 1
            elif state.process_goingWhere[pi] == +1:
                state.par_countGoingUp[npar] += 1
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


# ---------------------------------------------------------------------------------------------------------------------
# Observable routing and exchange statistics of a replica-exchange run: every measurement of every
# replica is accumulated under the CONTROL PARAMETER the replica holds at that moment, not under the rank that
# produced it.  Reference: ObservableHandlerPTCommon / ScalarObservableHandlerPT / VectorObservableHandlerPT
# (src/mpiobservablehandlerpt.h:171-212, src/mpiobservablehandlerpt.cpp:65-110, 176-215, 221-300) and
# DetQMCPT::saveReplicaExchangeStatistics / control_parameter_subdir (src/detqmcpt.h:596-660).
# ---------------------------------------------------------------------------------------------------------------------
def num_to_string(v):
    """numToString (src/tools.h:45-50): default ostream formatting of a number"""
    return ("%g" % v) if isinstance(v, float) else str(v)


def control_parameter_subdir(cpi, name, value):
    """src/detqmcpt.h:655-660"""
    return "p%d_%s%s" % (cpi, name, num_to_string(float(value)))


def _write_data_map(path, items, rows):
    """DataMapWriter::writeToFile (src/datamapwriter.h:116-160): header items in the order the reference adds them -- ("text", line)
    -> '## line', ("meta", map) -> '# key = value' sorted by key, ("kv", key, value) -> '# key = value' -- then
    key <tab> value [<tab> error] with 15 digits, scientific"""
    with open(path, "w") as f:
        for it in items:
            if it[0] == "text":
                f.write("## %s\n" % it[1])
            elif it[0] == "meta":
                for k in sorted(it[1]):
                    f.write("# %s = %s\n" % (k, it[1][k]))
            else:
                f.write("# %s = %s\n" % (it[1], it[2]))
        for row in rows:
            key = row[0]
            f.write((("%d" % key) if isinstance(key, (int, np.integer)) else ("%.15e" % key) if isinstance(key, float) else str(key)))
            for v in row[1:]:
                f.write("\t%.15e" % v)
            f.write("\n")


class ObservableRouterPT:
    """One handler for ALL observables of a replica-exchange run.  Collective: `insert` is called on every rank after a
    measurement sweep with the local replicas' current values; the values meet at rank 0 (one all_gather -- the
    reference does one MPI gather per observable) and are added, replica by replica, to the accumulators of the control
    parameter index that replica currently holds (`state.current_process_par`, kept up to date by replica_exchange_step).

    scalar_names / vector_specs = [(name, length)] fix the payload layout.  sweeps, jk_blocks, measure_interval as in
    DetQMCParams."""

    def __init__(self, state: ExchangeState, scalar_names, vector_specs, sweeps, jk_blocks=1, measure_interval=1, timeseries=False,
                 sweeps_has_changed=False):
        self.state = state
        self.scalar_names = list(scalar_names)
        self.vector_specs = [(n, int(l)) for n, l in vector_specs]
        self.width = len(self.scalar_names) + sum(l for _, l in self.vector_specs)
        self.sweeps, self.jk_blocks, self.measure_interval = int(sweeps), max(1, int(jk_blocks)), int(measure_interval)
        self.jk_block_size_sweeps = max(1, self.sweeps // self.jk_blocks)              # mpiobservablehandlerpt.h:141-142
        self.timeseries = timeseries
        self.sweeps_has_changed = bool(sweeps_has_changed)      # DetQMCParams::sweepsHasChanged: a resumed run with a new sweep count
        self.count_values = 0
        self.last_sweep_logged = 0
        self.nproc = len(state.controlParameterValues)
        self.is_root = bool(state.current_process_par)
        if self.is_root:
            self.par_total = np.zeros((self.nproc, self.width))
            self.par_jk = np.zeros((self.nproc, self.jk_blocks, self.width))
            self.par_series = [[] for _ in range(self.nproc)]

    def _pack(self, scalars, vectors):
        row = np.empty(self.width)
        row[:len(self.scalar_names)] = [float(scalars[n]) for n in self.scalar_names]
        o = len(self.scalar_names)
        for n, l in self.vector_specs:
            v = np.asarray(vectors[n], dtype=np.float64)
            if v.shape != (l,):
                raise ValueError("vector observable %s has shape %s, expected (%d,)" % (n, v.shape, l))
            row[o:o + l] = v
            o += l
        return row

    def insert(self, cur_sweep, local_values, dist, device="cpu"):
        """local_values: one (scalars dict, vectors dict) per local replica, in the order of the rank's replicas"""
        import torch
        nl = self.state.n_local
        if len(local_values) != nl:
            raise ValueError("expected %d local replicas, got %d" % (nl, len(local_values)))
        send = torch.from_numpy(np.stack([self._pack(s, v) for s, v in local_values])).to(device)
        if dist is not None:
            recv = [torch.empty_like(send) for _ in range(dist.get_world_size())]
            dist.all_gather(recv, send)
        else:
            recv = [send]
        if self.is_root:
            vals = np.concatenate([r.cpu().numpy() for r in recv], axis=0)          # [process][width]
            cur_block = cur_sweep // self.jk_block_size_sweeps                        # handleValues, :171-186
            for p_i in range(self.nproc):
                cpi = self.state.current_process_par[p_i]
                for jb in range(self.jk_blocks):
                    if jb != cur_block:
                        self.par_jk[cpi, jb] += vals[p_i]
                self.par_total[cpi] += vals[p_i]
                if self.timeseries:
                    self.par_series[cpi].append(vals[p_i, :len(self.scalar_names)].copy())
        self.count_values += 1
        self.last_sweep_logged = cur_sweep

    def evaluate_jackknife(self, cpi):
        """(mean, error) rows of width `width` for control parameter index cpi (evaluateJackknife, :188-212 and the
        scalar handler's plain standard deviation for a single block, mpiobservablehandlerpt.cpp:88-103)"""
        if not self.is_root:
            return None, None
        mean = self.par_total[cpi] / self.count_values
        err = np.zeros(self.width)
        if self.sweeps - self.last_sweep_logged <= self.measure_interval:
            # an error estimate needs several jackknife blocks, and blocks of a fixed size (mpiobservablehandlerpt.h:196)
            if self.jk_blocks > 1 and not self.sweeps_has_changed:
                block_samples = self.count_values // self.jk_blocks
                total_samples = self.count_values - block_samples
                blocks = self.par_jk[cpi] / total_samples
                bc = self.jk_blocks
                err = np.sqrt((bc - 1.0) / bc * np.sum((mean[None, :] - blocks) ** 2, axis=0))      # jackknife(), statistics.h:132-144
        # scalar observables with a single block: standard deviation of the buffered time series, variance() of
        # statistics.h:36-46 divides by N - 1 (mpiobservablehandlerpt.cpp:98-100; outside the end-of-run condition there too)
        if self.jk_blocks <= 1 and self.timeseries and len(self.par_series[cpi]) == self.count_values and self.count_values > 1:
            ns = len(self.scalar_names)
            ts = np.array(self.par_series[cpi])
            err[:ns] = np.sqrt(np.sum((ts - mean[None, :ns]) ** 2, axis=0) / (len(ts) - 1))
        return mean, err

    def write_results(self, directory, parameter_name, meta_model=None, meta_mc=None, meta_pt=None):
        """the reference's output tree: <directory>/p<cpi>_<name><value>/results.values and results-<vector>.values
        (outputResults, src/mpiobservablehandlerpt.cpp:221-300); rank 0 only"""
        import os
        if not self.is_root:
            return
        for cpi in range(self.nproc):
            value = self.state.controlParameterValues[cpi]
            sub = os.path.join(directory, control_parameter_subdir(cpi, parameter_name, value))
            os.makedirs(sub, exist_ok=True)
            mean, err = self.evaluate_jackknife(cpi)
            mm = dict(meta_model or {})
            mm[parameter_name] = num_to_string(float(value))                          # par_metaModel[cpi]: only that entry replaced
            meta = [("meta", mm), ("meta", dict(meta_mc or {})), ("meta", dict(meta_pt or {}))]
            ns = len(self.scalar_names)
            order = sorted(range(ns), key=lambda i: self.scalar_names[i])               # std::map<std::string, num>: sorted by name
            _write_data_map(os.path.join(sub, "results.values"),
                            [("text", "Monte Carlo results for observable expectation values")] + meta +
                            [("kv", "key", "observable"), ("text", "observable\t value \t error")],
                            [(self.scalar_names[i], mean[i], err[i]) for i in order])
            o = ns
            for name, l in self.vector_specs:                                           # :286-297
                _write_data_map(os.path.join(sub, "results-%s.values" % name),
                                [("text", "Monte Carlo results for vector observable %s expectation values" % name)] + meta +
                                [("kv", "key", "site"), ("kv", "observable", name), ("text", "key\t value \t error")],
                                [(float(i), mean[o + i], err[o + i]) for i in range(l)])
                o += l


def write_exchange_statistics(state: ExchangeState, directory, meta=None):
    """exchange-parameters.values, exchange-acceptance.values, exchange-diffusion.values
    (DetQMCPT::saveReplicaExchangeStatistics, src/detqmcpt.h:596-651); rank 0 only"""
    import os
    if not state.current_process_par:
        return
    n = len(state.controlParameterValues)
    metas = [("meta", dict(m)) for m in (meta or [])] if isinstance(meta, (list, tuple)) else [("meta", dict(meta or {}))]
    acc = [(state.par_swapUpAccepted[c] / state.par_swapUpProposed[c]) if state.par_swapUpProposed[c] else 0.0 for c in range(n)]
    df = []
    for c in range(n):
        up, down = state.par_countGoingUp[c], state.par_countGoingDown[c]
        df.append(up / (up + down) if (up + down) else 0.0)
    key = [("kv", "key", "control parameter index")]
    _write_data_map(os.path.join(directory, "exchange-parameters.values"),
                    metas + key + [("text", "Control parameter values"), ("text", "control parameter index \t control parameter value")],
                    [(c, float(state.controlParameterValues[c])) for c in range(n)])
    _write_data_map(os.path.join(directory, "exchange-acceptance.values"),
                    metas + key + [("text", "Acceptance ratio of exchanging replicas at control parameters (upwards)"),
                                   ("text", "control parameter index \t acceptance ratio")], [(c, acc[c]) for c in range(n)])
    _write_data_map(os.path.join(directory, "exchange-diffusion.values"),
                    metas + key + [("text", "Diffusion fraction of replicas at control parameters: df = nUp / (nUp + nDown)"),
                                   ("text", "control parameter index \t diffusion fraction")], [(c, df[c]) for c in range(n)])

# ---------------------------------------------------------------------------------------------------------------------
# The reference's configuration file and metadata headers, so that a run of scripts/run_pt.py on the reference's own
# simulation.conf writes the reference's own output tree (checked file by file against the tree the reference's
# detqmcptsdwo2 wrote: tests/golden/detqmcpt_run_o2_L4/expected, tests/test_gpu_parity.py).
# ---------------------------------------------------------------------------------------------------------------------
def parse_simulation_conf(path):
    """boost::program_options config-file syntax as the reference uses it (src/mpimaindetqmcptsdwopdim.cpp:60-262):
    `key = value` lines, `#` comments, a repeated key builds a list (rValues)."""
    conf = {}
    for raw in open(path):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        k, v = [t.strip() for t in line.split("=", 1)]
        if k in conf:
            if not isinstance(conf[k], list):
                conf[k] = [conf[k]]
            conf[k].append(v)
        else:
            conf[k] = v
    return conf


def _b(v):
    return str(v).strip().lower() in ("1", "true", "yes", "on")


def reference_metadata(conf, rvalues):
    """(model, mc, pt) metadata maps of a replica-exchange run as the reference prints them: prepareModelMetadataMap
    (src/detsdwopdim.cpp:363-438: numbers through numToString, the global-move switches as 0 / 1, the others as true / false),
    DetQMCParams::prepareMetadataMap (src/detqmcparams.cpp:60-98), DetQMCPTParams::prepareMetadataMap (src/detqmcptparams.cpp:40-68).
    Values are strings; `r` is filled in per control parameter by the caller."""
    g = lambda k, d: conf.get(k, d)
    f = lambda k, d: num_to_string(float(g(k, d)))
    tf = lambda k, d: "true" if _b(g(k, d)) else "false"
    i01 = lambda k, d: "1" if _b(g(k, d)) else "0"
    L = int(g("L", 4))
    m = int(g("m", 0)) or int(round(float(g("beta", 0)) / float(g("dtau", 0.1))))
    dtau = float(g("dtau", 0.1))
    model = {
        "model": "sdw", "opdim": str(int(g("opdim", 3))), "L": str(L), "N": str(L * L), "d": "2", "m": str(m), "s": str(int(g("s", 1))),
        "beta": num_to_string(m * dtau), "dtau": f("dtau", 0.1), "accRatio": f("accRatio", 0.5), "bc": g("bc", "pbc"), "c": f("c", 1.0),
        "cdwU": f("cdwU", 0.0), "checkerboard": tf("checkerboard", "false"), "delaySteps": str(int(g("delaySteps", 16))),
        "dumpGreensFunction": tf("dumpGreensFunction", "false"), "globalShift": i01("globalShift", "false"),
        "globalUpdateInterval": str(int(g("globalUpdateInterval", 100))), "lambda": f("lambda", 1.0), "mu": f("mu", 0.5),
        "overRelaxation": tf("overRelaxation", "false"), "phi2bosons": tf("phi2bosons", "false"), "phiFixed": tf("phiFixed", "false"),
        "repeatUpdateInSlice": str(int(g("repeatUpdateInSlice", 1))), "spinProposalMethod": g("spinProposalMethod", "box"),
        "turnoffFermionMeasurements": tf("turnoffFermionMeasurements", "false"), "turnoffFermions": tf("turnoffFermions", "false"),
        "txhor": f("txhor", -1.0), "txver": f("txver", -0.5), "tyhor": f("tyhor", 0.5), "tyver": f("tyver", 1.0), "u": f("u", 1.0),
        "updateMethod": g("updateMethod", "iterative"), "weakZflux": tf("weakZflux", "false"),
        "wolffClusterShiftUpdate": i01("wolffClusterShiftUpdate", "false"), "wolffClusterUpdate": i01("wolffClusterUpdate", "false"),
    }
    mi = int(g("measureInterval", 1))
    mc = {
        "greenUpdateType_string": g("greenUpdateType", "stabilized"), "jkBlocks": str(int(g("jkBlocks", 1))), "measureInterval": str(mi),
        "rngSeed": str(int(g("rngSeed", 0))), "saveConfigurationStreamBinary": tf("saveConfigurationStreamBinary", "false"),
        "saveConfigurationStreamInterval": str(int(g("saveConfigurationStreamInterval", mi))),
        "saveConfigurationStreamText": tf("saveConfigurationStreamText", "false"), "saveInterval": str(int(g("saveInterval", 0))),
        "simindex": str(int(g("simindex", 0))), "sweeps": str(int(g("sweeps", 0))), "thermalization": str(int(g("thermalization", 0))),
        "timeseries": tf("timeseries", "false"),
    }
    pt = {"controlParameterName": "r", "controlParameterValues": " ".join(num_to_string(float(v)) for v in rvalues),
          "exchangeInterval": str(int(g("exchangeInterval", 1)))}
    return model, mc, pt


def write_timeseries(router, directory, parameter_name, meta_model, meta_mc, meta_pt):
    """<obs>.series per control parameter (ScalarObservableHandlerPT::outputTimeseries, src/mpiobservablehandlerpt.cpp:105-160):
    one value per line at the stream's default precision; rank 0 only"""
    import os
    if not router.is_root or not router.timeseries:
        return
    for cpi in range(router.nproc):
        sub = os.path.join(directory, control_parameter_subdir(cpi, parameter_name, router.state.controlParameterValues[cpi]))
        os.makedirs(sub, exist_ok=True)
        mm = dict(meta_model or {})
        mm[parameter_name] = num_to_string(float(router.state.controlParameterValues[cpi]))
        for i, name in enumerate(router.scalar_names):
            with open(os.path.join(sub, name + ".series"), "w") as f:
                f.write("## Timeseries for observable %s\n" % name)
                for meta in (mm, meta_mc or {}, meta_pt or {}):
                    for k in sorted(meta):
                        f.write("# %s = %s\n" % (k, meta[k]))
                f.write("# observable = %s\n" % name)
                for row in router.par_series[cpi]:
                    f.write("%s\n" % num_to_string(float(row[i])))


def write_config_infoheader(directory, meta_model, meta_mc, meta_pt, cdw=False):
    """configs-phi.infoheader next to a configuration stream and, with cdwU != 0 (cdw = True), configs-l.infoheader next to the
    stream of the discrete field (src/detsdwopdim.cpp:5073-5108)"""
    import os
    files = [("configs-phi.infoheader", "## binary phi configuration stream (64 bit double precision floats) in file configs-phi.binarystream\n")]
    if cdw:
        files.append(("configs-l.infoheader", "## binary l configuration stream (32 bit signed integers) in file configs-l.binarystream\n"))
    for name, last in files:
        with open(os.path.join(directory, name), "w") as f:
            for meta in (meta_model, meta_mc, meta_pt):
                for k in sorted(meta):
                    f.write("#%s = %s\n" % (k, meta[k]))
            f.write(last)
