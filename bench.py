#!/usr/bin/env python3
"""Benchmark harness: DQMC sweeps/sec for the BASELINE.json headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Starts by itself for any N: the parent never touches a GPU, it spawns `--workers` worker processes per GPU
(`--device i`, default ONE), each holding `--batch` independent Markov chains spread over `--sub-batches` kernel contexts
that the host layer sweeps concurrently (one host thread + HIP stream per context; every launch of a context carries all
of its chains).  Under torch.distributed.run (WORLD_SIZE > 1, one rank per GPU) every
rank drives the workers of ITS GPU and the ranks meet at a barrier on both sides of the timed region; the result is the
same line.  A "step" is one sweepThermalization() of every chain.  value = all sweeps of all chains of all GPUs / the
slowest participant's time.  Workload: SDW O(2), L=16, beta=10, dtau=0.1 (m=100), s=10, checkerboard, delayed updates,
no fermion measurements.  Every chain is an independent Markov chain with its own RNG stream
(simindex), like the replicas the reference runs one per MPI process (`mpirun -n 8`); there is no data-path collective,
so scaling is weak.

The ONE JSON line carries, besides the contract's keys:
  per_gpu                    sweeps/s of every GPU
  one_process_sweeps_per_s   what ONE detsdw_create_batch handle in ONE process delivers (a DetQMCPT port owning one process per GPU)
  one_context_sweeps_per_s   ONE kernel context (one stream) alone on the GPU
  single_chain_sweeps_per_s  one context x one chain (latency of a single Markov chain)
  roofline                   the kernel family with the largest device time, measured with HIP events on the kernel's own
                             stream while ONE context has the GPU to itself (reproducible with rocprofv3, profiles/),
                             against its BINDING roof (HBM or fp64 MFMA, whichever fraction is larger)
  roofline_other_kernels, roofline_whole_step
  cpu_baseline, cpu_baseline_all_cores   the real reference binary (oracle/_ref) on this host, N = 1 only
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# delaySteps: depth of the delayed-update blocks -- a performance knob, the Markov chain does not depend on it (tests:
# test_update_slice_delay_steps_invariance, test_qr_mode_headline_size_vs_reference_checksums[32]); 32 halves the
# read-modify-write traffic of G per accepted update.  The reference CPU baseline runs with its own setting (16).
_COMMON = dict(dtau=0.1, s=10, r=-1.0, c=3.0, u=1.0, lambda_=1.0, mu=-0.5, txhor=-1.0, txver=-0.5, tyhor=0.5, tyver=1.0, bc="pbc", accRatio=0.5,
               rngSeed=1020304050,
               # same Green's functions and Markov chain as the reference-exact "svd" mode (tests), ~10x cheaper
               stabilisation=os.environ.get("DQMC_STABILISATION", "qr"))
# BASELINE.json configs 2-5 (config 1, the Hubbard replica, is CPU plumbing: tests/test_gpu_hubbard.py).  `--config` selects one; the
# default is the configuration the metric is quoted on (config 3).  batch / sub: chains per GPU and kernel contexts they are spread over.
CONFIGS = {
    "o2_L16_b10": dict(workload=dict(opdim=2, L=16, beta=10.0, delaySteps=32), batch=512, sub=4,
                       label="SDW-O2 L=16 beta=10", text="DetSDW O(2) L=16 beta=10 dtau=0.1 s=10"),
    "o2_L8_b5": dict(workload=dict(opdim=2, L=8, beta=5.0, delaySteps=32), batch=512, sub=4,
                     label="SDW-O2 L=8 beta=5", text="DetSDW O(2) L=8 beta=5 dtau=0.1 s=10"),
    "o2_L16_b20": dict(workload=dict(opdim=2, L=16, beta=20.0, delaySteps=32), batch=256, sub=4,
                       label="SDW-O2 L=16 beta=20", text="DetSDW O(2) L=16 beta=20 dtau=0.1 s=10"),
    # no magnetic flux: the reference refuses weakZflux for opdim = 3 (src/detsdwparams.cpp:57-60), and so does detsdw_create
    "o3_L24_b20": dict(workload=dict(opdim=3, L=24, beta=20.0, delaySteps=16), batch=8, sub=4,     # 8 chains in four contexts: 1.67 (two: 1.63, eight: 1.09; round 4, profiles/r04_bench_o3_L24_b20_pt8.json)
                       label="SDW-O3 L=24 beta=20", text="DetSDW O(3) L=24 beta=20 dtau=0.1 s=10 (no flux: the reference rejects O(3) + flux)",
                       ref_parts=True),
}
DEFAULT_CONFIG = "o2_L16_b10"
CONFIG = DEFAULT_CONFIG
WORKLOAD = {}


def select_config(name):
    global CONFIG, WORKLOAD
    CONFIG = name
    WORKLOAD = dict(_COMMON, **CONFIGS[name]["workload"])
    if os.environ.get("DQMC_DELAY_STEPS"):
        WORKLOAD["delaySteps"] = int(os.environ["DQMC_DELAY_STEPS"])
    if os.environ.get("DQMC_BENCH_PROPOSAL_BUDGET"):
        WORKLOAD["proposalBudget"] = int(os.environ["DQMC_BENCH_PROPOSAL_BUDGET"])
    # production variant (reference example/simulation.job:27-44): a global shift move every 10 sweeps
    if os.environ.get("DQMC_BENCH_GLOBAL_SHIFT"):
        WORKLOAD.update(globalShift=True, globalUpdateInterval=int(os.environ["DQMC_BENCH_GLOBAL_SHIFT"]))


select_config(DEFAULT_CONFIG)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F64_PEAK_TF = 78.6        # CDNA4 v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s (= fp64 vector rate)
# What the instruction sustains on the box with nothing else going on: scripts/micro/valu_mfma_mix.hip (8 independent accumulators per
# wave, 1-2 waves per SIMD, all operands in VGPRs), output kept under profiles/r03_valu_mfma_mix.log: 73.7-77.5 TFLOP/s = 0.94-0.99 of the
# guide's peak.  (Round 2 quoted 47.5 from scripts/micro/mfma_peak.hip; that kernel's loop carries eight v_accvgpr_read per MFMA -- the
# compiler moved its 12 accumulators between the two register files every iteration -- so it measured the copies, not the matrix pipe.)
# The kernels issue 3 real MFMAs per complex 16x16x4 step (3M), so their MFMA-instruction rate is 6/8 of the `achieved` figure, which
# counts 8 flop per complex multiply-add.
def _mfma_sustained():
    best = None
    try:
        for line in open(os.path.join(ROOT, "profiles", "r03_valu_mfma_mix.log")):
            if "v_mfma_f64_16x16x4 alone" in line and "MFMA group" in line:
                v = float(line.split("MFMA group")[1].split("TFLOP/s")[0])
                best = v if best is None else max(best, v)
    except Exception:
        pass
    return best


MFMA_F64_SUSTAINED_TF = _mfma_sustained()
DEFAULT_WORKERS = 1            # worker processes per GPU
FAKE = bool(os.environ.get("DQMC_BENCH_FAKE_WORKER"))     # CPU rehearsal of the control flow (tests/test_bench_cpu.py)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the real reference binary, 1 core and all cores
# ---------------------------------------------------------------------------------------------------------------------
def ref_exe():
    return os.path.join(ROOT, "oracle", "_ref", "ref_harness_fast_o%d" % WORKLOAD["opdim"])


def ref_args():
    return ["L=%d" % WORKLOAD["L"], "beta=%g" % WORKLOAD["beta"], "dtau=0.1", "s=10", "delaySteps=16", "opdim=%d" % WORKLOAD["opdim"]]


def ref_desc():
    return ("crstnbr/detqmc DetSDW<CB_ASSAAD_BERG,%d> built from the reference sources "
            "(-O3 -ffast-math -mavx2 -mfma, MKL 1 thread per process)" % WORKLOAD["opdim"])


def _ref_start(warmup, sweeps, tag):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    d = "/tmp/dqmc_ref_%d_%s" % (os.getpid(), tag)
    os.makedirs(d, exist_ok=True)
    return subprocess.Popen([ref_exe(), d] + ref_args() + ["mode=time", "warmup=%d" % warmup, "sweeps=%d" % sweeps], stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, text=True, env=env)


def cpu_baseline_parts(m, nst):
    """Sizes where ONE reference sweep takes hours (config 5: 40 SVDs of 2304 x 2304 per sweep): the reference binary on the SAME
    lattice at beta = 0.3 (3 time slices), its per-slice cost (local updates + wrap) and the cost of one stabilisation step
    (advanceDownGreen -> greenFromUdV) timed separately (mode=timeparts of oracle/ref_build/ref_harness.cpp) and extrapolated to
    sweep = m * slice + n * advance.  MKL gets all host cores here: a single core would need ten minutes for this sample."""
    ncpu = host_cores()
    env = dict(os.environ, MKL_NUM_THREADS=str(ncpu), OMP_NUM_THREADS=str(ncpu))
    d = "/tmp/dqmc_ref_%d_parts" % os.getpid()
    os.makedirs(d, exist_ok=True)
    args = [a for a in ref_args() if not a.startswith("beta=")] + ["beta=0.3", "mode=timeparts"]
    t0 = time.time()
    out = subprocess.run([ref_exe(), d] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, timeout=1500).stdout
    for line in out.splitlines():
        if line.startswith("REF_PARTS"):
            kv = dict(tok.split("=") for tok in line.split()[1:])
            ts, ta = float(kv["slice_seconds"]), float(kv["advance_seconds"])
            sweep_s = m * ts + nst * ta
            return {"value": 1.0 / sweep_s, "unit": "sweeps/s", "cores": ncpu, "kind": "reference",
                    "sample": ref_desc().replace("MKL 1 thread per process", "MKL %d threads" % ncpu) +
                              ": same lattice at beta = 0.3 (n_g = %s), %.2f s per time slice (local updates + wrap) and %.1f s per "
                              "stabilisation step (advanceDownGreen) measured in %.0f s of wall time incl. construction, extrapolated to "
                              "m = %d slices + n = %d steps = %.0f s per sweep" % (kv["ng"], ts, ta, time.time() - t0, m, nst, sweep_s)}
    raise RuntimeError("reference harness printed no REF_PARTS line")


def _ref_finish(p, timeout):
    out, _ = p.communicate(timeout=timeout)
    for line in out.splitlines():
        if line.startswith("REF_TIMING"):
            kv = dict(tok.split("=") for tok in line.split()[1:])
            return int(kv["sweeps"]), float(kv["seconds"])
    raise RuntimeError("reference harness printed no REF_TIMING line")


def cpu_baseline_port():
    """fallback when the reference binary cannot run on this host: the numpy oracle, bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))   # cpu_baseline leg: the oracle is what is timed here
    from detsdw_oracle import DetSDWOracle, SDWParams as OP
    kw = {k: v for k, v in WORKLOAD.items() if k in OP.__dataclass_fields__ and k != "stabilisation"}
    o = DetSDWOracle(OP(**kw))
    m, n, s_ = o.m, o.n, o.s
    t0 = time.time()
    k, nsl = m, 0
    while nsl < (m - (n - 1) * s_) and (time.time() - t0 < 20.0 or nsl == 0):
        o.updateInSliceThermalization(k)
        o.wrapDownGreen(k)
        k -= 1
        nsl += 1
    sweep_s = m * (time.time() - t0) / nsl
    sample = "%d of %d time slices (local updates + wrap) in %.1f s" % (nsl, m, time.time() - t0)
    if k == (n - 1) * s_:
        t1 = time.time()
        o.advanceDownGreen(n)
        sweep_s += n * (time.time() - t1)
        sample += ", one stabilisation step in %.1f s" % (time.time() - t1)
    else:
        sample += ", stabilisation steps not timed (lower bound of the sweep time)"
    import threadpoolctl
    thr = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1])
    return {"value": 1.0 / sweep_s, "unit": "sweeps/s", "cores": thr, "kind": "port",
            "sample": "oracle/detsdw_oracle.py (numpy + scipy zgesvd), extrapolated to a full sweep from " + sample}


def host_cores():
    """cores this job may use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 logical CPUs and
    grants 16 cores' worth of time)"""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            ncpu = min(ncpu, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return ncpu


class CpuBaseline:
    """1 core: ONE reference process, 1 warm-up + `sweeps` timed sweepThermalization(); may run while the GPU post-phases
    (one busy host thread) are in progress -- the box has 16+ cores.  All cores: one reference process per core at the same
    time (how the reference scales: independent replicas under mpirun), aggregate sweeps/s."""

    def __init__(self, sweeps=2):
        self.sweeps = sweeps
        self.p1 = None
        self.parts = bool(CONFIGS[CONFIG].get("ref_parts"))
        if self.parts:
            return
        if os.path.exists(ref_exe()) and not FAKE:
            try:
                self.p1 = _ref_start(1, sweeps, "one")
            except Exception as e:
                sys.stderr.write("reference binary unusable (%r)\n" % (e,))

    def finish(self, m=None, nst=None):
        out = {}
        if self.parts:
            try:
                out["cpu_baseline"] = cpu_baseline_parts(m, nst)
            except Exception as e:
                sys.stderr.write("reference timeparts run failed (%r)\n" % (e,))
                out["cpu_baseline"] = None
            return out
        if self.p1 is None:
            out["cpu_baseline"] = cpu_baseline_port()
            return out
        try:
            n, sec = _ref_finish(self.p1, 600)
        except Exception as e:
            sys.stderr.write("reference binary failed (%r), timing the oracle port instead\n" % (e,))
            out["cpu_baseline"] = cpu_baseline_port()
            return out
        out["cpu_baseline"] = {"value": n / sec, "unit": "sweeps/s", "cores": 1, "kind": "reference",
                               "sample": ref_desc() + ": %d sweepThermalization() after init + 1 warm-up sweep, %.1f s" % (n, sec)}
        ncpu = host_cores()
        try:
            ps = [_ref_start(1, self.sweeps, "all%d" % i) for i in range(ncpu)]
            res = [_ref_finish(p, 900) for p in ps]
            out["cpu_baseline_all_cores"] = {
                "value": sum(n_ / s_ for n_, s_ in res), "unit": "sweeps/s", "cores": ncpu, "kind": "reference",
                "sample": ref_desc() + ": %d independent replicas at the same time (one process per core, as under mpirun), each %d "
                                     "sweepThermalization() after init + 1 warm-up sweep; slowest %.1f s, fastest %.1f s"
                                     % (ncpu, self.sweeps, max(s_ for _, s_ in res), min(s_ for _, s_ in res))}
        except Exception as e:
            sys.stderr.write("all-cores baseline failed (%r)\n" % (e,))
        return out


# ---------------------------------------------------------------------------------------------------------------------
# worker: one kernel context
# ---------------------------------------------------------------------------------------------------------------------
class _FakeBatch:
    """stands in for DetSDWBatch in the CPU rehearsal of the control flow (no GPU, no library)"""

    def __init__(self, B, device=0):
        self.B, self.device, self.timed = B, device, False

    def sweepThermalization(self):
        time.sleep(0.01)
        # rehearsal of a worker that dies mid-run (tests/test_bench_cpu.py): DQMC_BENCH_FAKE_DIE = "<device>:<exit code>"
        die = os.environ.get("DQMC_BENCH_FAKE_DIE")
        if die and self.timed and int(die.split(":")[0]) == self.device:
            os._exit(int(die.split(":")[1]))


def worker(a, readline=None, emit=None):
    """build, warm up, wait for GO, run the timed steps, report; then optionally repeat ALONE with profiling on"""
    import dataclasses
    readline = readline or sys.stdin.readline
    emit = emit or (lambda line: print(line, flush=True))
    B = max(1, a.batch)
    if FAKE:
        batch, ctx = _FakeBatch(B, a.device), None
    else:
        from detqmc_amd import DetSDWBatch, SDWParams
        p0 = SDWParams(device=a.device, **WORKLOAD)
        if a.exchange:
            # the chains of this worker are the replicas of ONE parallel-tempering ensemble in r (BASELINE configs 4 / 5: 8 replicas):
            # a ladder of control parameters, replicaExchangeStep after every sweep (detqmc_amd/pt.py; src/detqmcpt.h:963-1118)
            rvals = [-1.4 + 0.8 * i / max(B - 1, 1) for i in range(B)]
            batch = DetSDWBatch([dataclasses.replace(p0, simindex=a.simindex * B + i, r=rvals[i]) for i in range(B)], sub_batches=a.sub_batches)
        else:
            batch = DetSDWBatch([dataclasses.replace(p0, simindex=a.simindex * B + i) for i in range(B)], sub_batches=a.sub_batches)
        ctx = batch.kernel_context
    if a.exchange and not FAKE:
        from detqmc_amd.pt import ExchangeState, ReplicaAdapter, replica_exchange_step
        pt_reps = [ReplicaAdapter(batch.chain(b)) for b in range(B)]
        pt_state = ExchangeState.create(rvals, 0, 1, B)
        sweep_only = batch.sweepThermalization

        def sweep_and_exchange():
            sweep_only()
            replica_exchange_step(pt_reps, pt_state, None)
        batch.sweepThermalization = sweep_and_exchange
    for _ in range(a.warmup):
        batch.sweepThermalization()
    if ctx:
        ctx.synchronize()
    emit("READY")
    if readline().strip() != "GO":
        return
    if FAKE:
        batch.timed = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batch.sweepThermalization()          # one C call per lockstep sweep of all B chains
    if ctx:
        ctx.synchronize()
    dt = time.perf_counter() - t0
    if FAKE:
        out = {"dt": dt, "n_g": 512, "m": 100, "acceptance": [0.5]}
    else:
        info = batch.chain(0).info
        out = {"dt": dt, "n_g": info.n_g, "m": info.m,
               "acceptance": [batch.chain(i).info.lastAccRatioLocal_phi for i in range(min(B, 4))]}
    emit("RESULT " + json.dumps(out))
    # then, if asked: the same steps again with per-kernel HIP-event timing
    #   SOLO  -- this worker's (first) context, for the profile worker that runs ONE context alone on the GPU
    #   PROF4 -- ALL contexts of this worker at once, each with events on its own stream: the per-family device times in the regime
    #            `value` is measured in (the contexts overlap, so a family's time includes waiting for the other contexts' kernels)
    while True:
        cmd = readline().strip()
        if cmd == "SOLO":
            if FAKE:
                emit("SOLO " + json.dumps({"dt_profiled": dt, "prof": None}))
                break
            ctx.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                batch.sweepThermalization()
            ctx.synchronize()
            dts = time.perf_counter() - t0
            prof = ctx.profile_read()
            ctx.profile_enable(False)
            emit("SOLO " + json.dumps({"dt_profiled": dts, "prof": {k: list(v) if isinstance(v, tuple) else v for k, v in prof.items()}}))
        elif cmd == "PROF4":
            if FAKE:
                emit("PROF4 " + json.dumps({"dt_profiled": dt, "contexts": 0, "families": None}))
                continue
            ctxs = batch.kernel_contexts()
            for c in ctxs:
                c.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                batch.sweepThermalization()
            for c in ctxs:
                c.synchronize()
            dts = time.perf_counter() - t0
            fams = {}
            for c in ctxs:
                pr = c.profile_read()
                c.profile_enable(False)
                for k, v in pr.items():
                    if isinstance(v, tuple) and k != "jacobi":
                        cur = fams.setdefault(k, [0.0, 0])
                        cur[0] += v[0]; cur[1] += v[1]
                cur = fams.setdefault("qr_apply", [0.0, 0])
                cur[0] += pr["decomp_round_ms"]; cur[1] += pr["decomp_rounds"]
                for k, v in pr.get("sub", {}).items():
                    cur = fams.setdefault(k, [0.0, 0])
                    cur[0] += v[0]; cur[1] += v[1]
            emit("PROF4 " + json.dumps({"dt_profiled": dts, "contexts": len(ctxs), "families": fams}))
        else:
            break
    if not FAKE:
        batch.close()


class InprocWorker:
    """The worker protocol inside this process (one context, no child processes): what `rocprofv3 -- python3 bench.py
    --inprocess` can profile -- on this pool a profiled process must not spawn GPU children."""

    def __init__(self, a):
        import copy
        import queue
        import threading
        self.inq, self.outq = queue.Queue(), queue.Queue()
        self.stdin, self.stdout = self, self
        self.th = threading.Thread(target=worker, args=(copy.copy(a), self.inq.get, self.outq.put), daemon=True)
        self.th.start()

    def write(self, line):
        self.inq.put(line)

    def flush(self):
        pass

    def readline(self):
        while self.th.is_alive() or not self.outq.empty():
            try:
                return self.outq.get(timeout=0.5)
            except Exception:
                continue
        return ""

    def poll(self):
        return None if self.th.is_alive() else 0

    def wait(self):
        self.th.join()


# ---------------------------------------------------------------------------------------------------------------------
# rooflines from the per-family HIP-event times of ONE context alone on the GPU
# ---------------------------------------------------------------------------------------------------------------------
def qr_apply_work(nn):
    """algorithmic bytes and flops of the k_qr_apply_reg launches of ONE factorisation incl. explicit Q (run_qr,
    kernels_qr.hip): a launch reads and writes its columns once and, per block reflector, does two rows x 16 x ncols
    complex products (W = V^H C, C += V W2)."""
    byt = flo = 0.0
    p, npan = 0, (nn + 15) // 16

    def add(rows, ncols, nref):
        nonlocal byt, flo
        byt += 2 * 16.0 * rows * ncols + nref * 16.0 * rows * 16
        flo += nref * 2 * 8.0 * rows * 16 * ncols

    while p < npan:
        left = nn - p * 16
        if nn <= 512 and left > 64:
            for rows, ncols, nref in ((left, 16, 1), (left, 32, 2), (left - 32, 16, 1), (left, left - 64, 4)):
                add(rows, ncols, nref)
            p += 4
        elif nn <= 512 and left > 32:
            add(left, 16, 1)
            add(left, left - 32, 2)
            p += 2
        else:
            nbp = min(16, left)
            if left - nbp > 0:
                add(left, left - nbp, 1)
            p += 1
    p = npan - 1
    while p >= 0:                       # formation of Q, reflectors in reverse order
        cnt = 1
        if nn <= 512 and min(16, nn - p * 16) == 16:
            cnt = 4 if p >= 3 else (2 if p >= 1 else 1)
        lo = p - (cnt - 1)
        add(nn - lo * 16, nn - lo * 16, cnt)
        p -= cnt
    return byt, flo


def rooflines(rawprof, n, m, B, traffic):
    """one entry per kernel family: achieved algorithmic GB/s and TFLOP/s over the family's device time, the fraction of
    each peak, and `bound` = the roof it is closer to.  `traffic` = HBM bytes per launch from the PMC passes, if taken."""
    prof = {k: (tuple(v) if isinstance(v, list) else v) for k, v in rawprof.items()}
    if isinstance(prof.get("sub"), dict):
        prof["sub"] = {k: list(v) for k, v in prof["sub"].items()}
    OPD = WORKLOAD["opdim"]
    MSF = 4 if OPD == 3 else 2
    N, D, s = n // MSF, WORKLOAD["delaySteps"], WORKLOAD["s"]

    def entry(name, kernel, ms, launches, total_bytes, total_flops, note, latency_bound=False):
        gbs = total_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tfs = total_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        fh, fm = gbs / HBM_PEAK_GBS, tfs / MFMA_F64_PEAK_TF
        mf = fm > fh
        e = {"family": name, "kernel": kernel, "bound": "mfma" if mf else "hbm",
             "achieved": tfs if mf else gbs, "peak": MFMA_F64_PEAK_TF if mf else HBM_PEAK_GBS, "unit": "TFLOP/s" if mf else "GB/s",
             "frac": fm if mf else fh, "traffic": None,
             "hbm_GBps": gbs, "hbm_frac": fh, "mfma_TFLOPps": tfs, "mfma_frac": fm,
             "algorithmic_bytes_per_launch": total_bytes / max(launches, 1), "algorithmic_flops_per_launch": total_flops / max(launches, 1),
             "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": launches, "device_ms": ms, "chains_per_launch": B, "note": note}
        if latency_bound:
            e["latency_bound"] = True
        if mf:
            e["mfma_instruction_TFLOPps"] = 0.75 * tfs                      # 3M: 6 of the 8 counted flop are issued
            e["mfma_instruction_frac_of_peak"] = 0.75 * tfs / MFMA_F64_PEAK_TF
        t = (traffic or {}).get(name)
        if t:
            # HBM bytes per launch from the PMC passes (their own run of this configuration) and, next to it, that run's
            # own algorithmic bytes per launch: for the update kernels both depend on the acceptance of the run
            e["traffic"] = t["hbm_bytes_per_launch"]
            e["traffic_over_algorithmic"] = t.get("traffic_over_algorithmic")
            e["traffic_note"] = t.get("note", "")
        return e

    roofs = []
    blocks = max(prof["blocks_nonempty"], 1)          # (chain, block) pairs that flushed
    acc = prof["updates_accepted"]                     # accepted updates, all chains
    # decision kernel: per proposal (OPDIM+1) uniforms, 2*OPDIM neighbouring-slice field values, cosh/sinh, G[c,I], G[I,c],
    # G[c,c], G[c,prev], G[prev,c] with |I| = MSF*D/2 on average
    cand_bytes = (OPD + 1) * 8 + 2 * OPD * 8 + 16 + (2 * MSF * (MSF * D // 2) + 3 * MSF * MSF) * 16
    nslices = prof["decide"][1] // ((N + D - 1) // D)
    roofs.append(entry("decide", "k_update_decide<%d>" % OPD, *prof["decide"], float(N) * cand_bytes * nslices * B, 0.0,
                       "sequential Metropolis chain of one slice: ONE workgroup per chain by construction -- latency bound, "
                       "neither roof is its limit; launches of finished slices exit at once", latency_bound=True))
    if WORKLOAD["stabilisation"] == "svd":
        roofs.append(entry("decomp", "k_jacobi_round<8,2>", prof["decomp_round_ms"], prof["decomp_rounds"],
                           4.0 * n * n * 16.0 * B * prof["decomp_rounds"], 0.0, "one Jacobi round reads and writes every column of A and V once"))
    else:
        qb, qf = qr_apply_work(n)
        calls = max(prof["qr_calls"], 1)
        # qr_calls = chain (UDT) factorisations: factor + explicit Q.  (A Green's function that still goes through the Householder
        # route, n_g > 512, applies Q^H to a full matrix instead of forming Q: same launch shapes.)
        roofs.append(entry("qr_apply", "k_qr_apply_reg", prof["decomp_round_ms"], max(prof["decomp_rounds"], 1), qb * calls * B, qf * calls * B,
                           "block reflectors of up to 4 panels applied to the trailing matrix / to Q with the columns in registers: "
                           "one read + one write per launch, 2 x 8 rows 16 ncols flop per reflector"))
        sub = prof.get("sub") or {}
        sub_ms = 0.0
        for key, kern, note in (("lu_update", "k_flush<..., TAG = 1> / k_flush_lds<1>", "trailing updates of the LU factorisation inside greenFromUdV (K = 32): a read-modify-write "
                                 "stream over the trailing matrix with a thin product, on the flush kernel"),
                                ("fact_gemm", "k_zgemm<..., TAG = 1>", "products inside factorisations: the levels of the recursive triangular solves (N = K = 32 ... 256), "
                                 "the block Gram-Schmidt QR for n_g > 1024")):
            v = sub.get(key)
            if v and v[1] > 0:
                roofs.append(entry(key, kern, v[0], v[1], v[3], v[2], note))
                sub_ms += v[0]
        rest_ms = max(prof["decomp"][0] - prof["decomp_round_ms"] - sub_ms, 0.0)
        roofs.append(entry("qr_rest", "k_qr_panel, " + ("LU of the Green's function (k_lu_panel, k_lu_rowswap_trsm, K = 32 updates), " if n <= 512 else "") + "triangular solves, glue", rest_ms, max(prof["decomp"][1] - prof["decomp_rounds"] - sum(int(v[1]) for v in sub.values()), 1),
                           0.0, 0.0, "panel factorisations (a chain of dependent reductions: latency bound) and the small kernels "
                           "around the QR; no roofline claimed", latency_bound=True))
    bl = prof["bmult"][1]
    nst = (m + s - 1) // s
    slices_per_launch = (2.0 * m + nst * s) / (2.0 * m + nst)      # per sweep: 2 m single-slice wraps + n chains of s slices
    roofs.append(entry("bmult", "k_bmult_chain", *prof["bmult"], 2 * 16.0 * n * n * B * bl, 56.0 * n * n * B * bl * slices_per_launch,
                       "one read + one write of A per chain of slices; ~56 flop per element and slice on the vector ALU"))
    roofs.append(entry("gather", "k_update_gather", *prof["gather"], 4 * 16.0 * n * MSF * acc, 8.0 * n * MSF * MSF * acc * acc / blocks,
                       "X = G[:,I] W and Gr = G[I,:] - E: reads the accepted rows/columns of G, writes the flush operands; "
                       "blocks without work exit at once"))
    roofs.append(entry("flush", "k_flush (G += X Gr)", *prof["flush"], 2 * 16.0 * n * n * blocks + 2 * 16.0 * n * MSF * acc, 8.0 * n * n * MSF * acc,
                       "read-modify-write of G once per delayed-update block and chain that accepted an update (+ the operand "
                       "panels), 8 n_g^2 MSF flop per accepted update on the matrix cores"))
    gms, gl = prof["gemm"]
    roofs.append(entry("gemm", "k_zgemm<2,2>", gms, gl, 3 * 16.0 * n * n * B * gl, prof["gemm_flops"],
                       "n_g^3 complex products on v_mfma_f64_16x16x4_f64, 3 real MFMAs per complex step (3M), flops counted as 8 M N K; bytes: A and B read, C written once"))
    # `roofline` = the family with the largest device time among the kernels that fill the chip on their own (those a roof
    # can bound); the latency-bound ones (one workgroup per chain: decision kernel, QR panel) follow in the list with their
    # device time, flagged, and no roofline is claimed for them
    roofs.sort(key=lambda r: (bool(r.get("latency_bound")), -r["device_ms"]))
    tot_ms = sum(r["device_ms"] for r in roofs)
    tot_b = sum(r["algorithmic_bytes_per_launch"] * r["launches"] for r in roofs if not r.get("latency_bound"))
    tot_f = sum(r["algorithmic_flops_per_launch"] * r["launches"] for r in roofs)
    whole = {"device_ms": tot_ms, "hbm_GBps": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0,
             "mfma_TFLOPps": tot_f / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0}
    whole["hbm_frac"] = whole["hbm_GBps"] / HBM_PEAK_GBS
    whole["mfma_frac"] = whole["mfma_TFLOPps"] / MFMA_F64_PEAK_TF
    whole["note"] = "algorithmic bytes and flops of all families / summed device time of one context alone on the GPU"
    return prof, roofs, whole


def load_traffic(B, D):
    """HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes, WRITE_SIZE),
    taken with scripts/pmc_collect.sh + scripts/pmc_traffic_summary.py for this batch size and delay depth"""
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic_b%d_d%d.json" % (rnd, B, D))
        if os.path.exists(path):
            try:
                return json.load(open(path))["families"]
            except Exception:
                pass
    return None


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS),
                    help="BASELINE.json workload: o2_L16_b10 (config 3, the one the metric is quoted on; default), o2_L8_b5 (2), "
                         "o2_L16_b20 (4), o3_L24_b20 (5, without the flux the reference rejects)")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("DQMC_WORKERS_PER_GPU", str(DEFAULT_WORKERS))),
                    help="worker processes (kernel contexts) per GPU")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DQMC_CHAINS_PER_WORKER", "0")),
                    help="independent Markov chains per worker process (default: the configuration's)")
    ap.add_argument("--sub-batches", type=int, default=int(os.environ.get("DQMC_SUB_BATCHES", "0")),
                    help="kernel contexts per worker process the chains are spread over (swept concurrently, one host thread each)")
    ap.add_argument("--exchange", action="store_true",
                    help="the chains of a worker are the replicas of one parallel-tempering ensemble (r ladder -1.4 ... -0.6): a replica-exchange "
                         "step after every sweep, inside the timed region (BASELINE configs 4 / 5: --batch 8)")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE context in this process instead of worker processes (for rocprofv3)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--simindex", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args()
    select_config(a.config)
    if a.batch <= 0:
        a.batch = CONFIGS[a.config]["batch"]
    if a.sub_batches <= 0:
        a.sub_batches = CONFIGS[a.config]["sub"]
    # small contexts (8 replicas of an exchange ensemble, a single chain): shallower delay blocks are faster there -- 8 chains of config 4
    # run 47.2 sweeps/s at depth 16 against 44.2 at 32 (profiles/r04_small_batch_delay_scan.log); the chain does not depend on the depth
    if not os.environ.get("DQMC_DELAY_STEPS") and a.batch // max(1, a.sub_batches) <= 32:
        WORKLOAD["delaySteps"] = min(WORKLOAD["delaySteps"], 16)
    if a.worker:
        return worker(a)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = bool(os.environ.get("DQMC_BENCH_ONE_DEVICE"))     # rehearsal on a one-GPU box: every "GPU" is device 0
    dist = torch = None
    backend = os.environ.get("DQMC_BENCH_BACKEND", "nccl")
    if world > 1:
        # launched by torch.distributed.run, one rank per GPU: the ranks only meet at the barriers around the timed region
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(0 if one_device else local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", 0 if one_device else local))
        else:
            dist.init_process_group(backend)
        n_gpus, my_gpus = world, [local]
    else:
        # started as plain `python bench.py --gpus N`: this process drives the workers of all N GPUs itself
        n_gpus, my_gpus = max(1, a.gpus), list(range(max(1, a.gpus)))

    def fence():
        if dist is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    R = 1 if a.inprocess else max(1, a.workers)
    B = max(1, a.batch)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)

    S = max(1, a.sub_batches)
    while B % S != 0:          # the contexts of a process hold the same number of chains
        S -= 1

    def spawn(device, simindex, batch, steps, warmup, sub=None, extra_env=None):
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--config", a.config, "--device", str(0 if one_device else device), "--simindex",
               str(simindex), "--steps", str(steps), "--warmup", str(warmup), "--batch", str(batch), "--sub-batches", str(sub or S)] + (["--exchange"] if a.exchange else [])
        return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=dict(env, **(extra_env or {})))

    procs = []          # (gpu index, process)
    if a.inprocess:
        a.device, a.simindex = (0 if one_device else local), rank
        procs.append((my_gpus[0], InprocWorker(a)))
    else:
        for g in my_gpus:
            gi = g if world == 1 else rank          # global GPU index -> distinct RNG streams on every GPU
            for i in range(R):
                procs.append((gi, spawn(g, gi * R + i, B, a.steps, a.warmup)))

    def read_tag(p, tag):
        while True:
            line = p.stdout.readline()
            if not line:
                # a dead worker ends the whole job with a non-zero status that names it: no JSON line is printed
                rc = p.wait() if hasattr(p, "wait") else p.poll()
                for _, q in procs:
                    if q is not p and q.poll() is None and hasattr(q, "kill"):
                        q.kill()
                sys.stderr.write("bench worker died while the parent waited for %s (exit code %s)\n" % (tag, rc))
                raise SystemExit(3)
            if line.startswith(tag):
                return line[len(tag):].strip()

    def send(p, word):
        p.stdin.write(word + "\n")
        p.stdin.flush()

    for _, p in procs:
        read_tag(p, "READY")            # replicas built, warm-up sweeps done, devices idle
    fence()
    t0 = time.perf_counter()
    for _, p in procs:
        send(p, "GO")
    results = [json.loads(read_tag(p, "RESULT")) for _, p in procs]    # each worker synchronised its stream
    fence()
    dt = time.perf_counter() - t0
    # per-GPU rate: the chains of a GPU over the time of its slowest context
    per_gpu = {}
    for (g, _), r in zip(procs, results):
        per_gpu[g] = max(per_gpu.get(g, 0.0), r["dt"])
    per_gpu = [R * B * a.steps / per_gpu[g] for g in sorted(per_gpu)]

    # post-phases (rank 0, first GPU), after every timed worker has left the GPU:
    #   ONE kernel context (B / S chains, one stream) alone: its rate, then the same steps with per-kernel HIP-event timing;
    #   ONE context with ONE chain: the latency of a single Markov chain
    lead = rank == 0
    inproc_solo = None
    prof4 = None
    if lead and not a.inprocess and R == 1 and n_gpus == 1:
        # the timed regime once more with per-family event timing in every context (only when ONE worker drives the whole GPU)
        send(procs[0][1], "PROF4")
        prof4 = json.loads(read_tag(procs[0][1], "PROF4"))
    for i, (_, p) in enumerate(procs):
        send(p, "SOLO" if (a.inprocess and lead and i == 0) else "QUIT")
    if a.inprocess and lead:
        inproc_solo = json.loads(read_tag(procs[0][1], "SOLO"))
        send(procs[0][1], "QUIT")
    for _, p in procs:
        p.wait()
    cpu = CpuBaseline() if (lead and n_gpus == 1 and not a.no_cpu_baseline) else None     # 1-core run overlaps the post-phases
    single = one_ctx = solo = None
    Bc = B // S                                                                          # chains of one kernel context
    if lead and not a.inprocess:
        pw = spawn(my_gpus[0], 7777, Bc, a.steps, a.warmup, sub=1)
        read_tag(pw, "READY")
        send(pw, "GO")
        one_ctx = Bc * a.steps / json.loads(read_tag(pw, "RESULT"))["dt"]
        send(pw, "SOLO")
        solo = json.loads(read_tag(pw, "SOLO"))
        send(pw, "QUIT")
        pw.wait()
        nst = max(2, min(a.steps, 4))
        # one context x ONE chain.  The delay depth is a performance knob (same chain for every depth): a batch amortises the flush of
        # a deep block over many chains, a single chain is bound by the decisions, whose cost grows with the block -- 16 is its
        # optimum at n_g = 512 (6.3 / 6.7 / 6.8 / 6.8 / 6.7 / 6.3 sweeps/s at 8 / 12 / 16 / 20 / 24 / 32, gpurun call 33 of round 3)
        single_D = min(WORKLOAD["delaySteps"], 16)
        sp = spawn(my_gpus[0], 9999, 1, nst, 1, sub=1, extra_env=None if os.environ.get("DQMC_DELAY_STEPS") else {"DQMC_DELAY_STEPS": str(single_D)})
        read_tag(sp, "READY")
        send(sp, "GO")
        sr = json.loads(read_tag(sp, "RESULT"))
        send(sp, "QUIT")
        sp.wait()
        single = nst / sr["dt"]
    elif lead:
        solo, one_ctx = inproc_solo, B * a.steps / results[0]["dt"]
        Bc = B
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allg = [None] * world
        dist.all_gather_object(allg, per_gpu)
        per_gpu = [v for part in allg for v in part]

    if lead:
        r0 = results[0]
        n = r0["n_g"]
        res = {
            "metric": "DQMC sweeps/sec (%s fp64)" % CONFIGS[a.config]["label"],
            "value": n_gpus * R * B * a.steps / dt,
            "unit": "sweeps/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "sweeps_per_s_per_chain": a.steps / dt,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (random initial field, fixed seed)",
            "config": {"workload": CONFIGS[a.config]["text"] + " checkerboard delayed(%d) " % WORKLOAD["delaySteps"] +
                                   "sweepThermalization, %d independent chains per GPU (%d process x %d kernel contexts x %d lockstep chains), stabilisation=%s%s"
                                   % (R * B, R, S, B // S, WORKLOAD["stabilisation"],
                                      (", global shift move every %d sweeps" % WORKLOAD["globalUpdateInterval"] if WORKLOAD.get("globalShift") else "") +
                                      (", the chains of a process are ONE replica-exchange ensemble in r (ladder -1.4 ... -0.6), exchange step after every sweep" if a.exchange else "")),
                       "n_g": n, "m": r0["m"], "replicas_per_gpu": R * B, "processes_per_gpu": R, "contexts_per_process": S,
                       "chains_per_context": B // S,
                       "launch": "torch.distributed.run" if world > 1 else "self"},
            "per_gpu": per_gpu,
            # what ONE detsdw_create_batch handle (one process) delivers: the headline itself when processes_per_gpu is 1
            "one_process_sweeps_per_s": per_gpu[0] / R,
            # ONE kernel context (dqmc_create_batch: one stream, one launch sequence) of chains_per_context chains alone on the GPU
            "one_context_sweeps_per_s": one_ctx,
            "single_chain_sweeps_per_s": single,
            "single_chain_delaySteps": (int(os.environ["DQMC_DELAY_STEPS"]) if os.environ.get("DQMC_DELAY_STEPS") else min(WORKLOAD["delaySteps"], 16)),
            "acceptance": r0["acceptance"],
        }
        if solo and solo.get("prof"):
            prof, roofs, whole = rooflines(solo["prof"], n, r0["m"], Bc, load_traffic(Bc, WORKLOAD["delaySteps"]))
            res["roofline"] = roofs[0]
            res["roofline_other_kernels"] = roofs[1:]
            res["roofline_whole_step"] = whole
            res["roofline_selection"] = ("largest device time among the kernel families a roof can bound; the decision kernel (%.0f ms) and the QR "
                                         "panel / LU / glue kernels (%.0f ms) are latency bound (one workgroup per chain) and listed under "
                                         "roofline_other_kernels with latency_bound = true" % (
                                             sum(r["device_ms"] for r in roofs if r["family"] == "decide"),
                                             sum(r["device_ms"] for r in roofs if r["family"] == "qr_rest")))
            res["roofline_conditions"] = ("HIP events on the context's own stream, ONE context (%d chains) alone on the GPU, %d steps right "
                                          "after the timed region (%.1f sweeps/s with the event records)" % (Bc, a.steps, Bc * a.steps / solo["dt_profiled"]))
            if MFMA_F64_SUSTAINED_TF:
                res["mfma_f64_sustained"] = {"TFLOPps": MFMA_F64_SUSTAINED_TF, "frac_of_peak": MFMA_F64_SUSTAINED_TF / MFMA_F64_PEAK_TF,
                                             "note": "v_mfma_f64_16x16x4_f64 issued back to back from VGPR accumulators, 1-2 waves per SIMD, measured on this "
                                                     "pool's MI355X (scripts/micro/valu_mfma_mix.hip, profiles/r03_valu_mfma_mix.log): the guide's peak is reachable"}
            res["device_ms_by_family"] = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items()
                                          if isinstance(v, tuple) and k != "jacobi"}
            res["decompositions"] = {"svd_calls": prof["svd_calls"], "jacobi_sweeps": prof["svd_sweeps_total"],
                                     "max_sweeps": prof["svd_sweeps_max"], "qr_calls": prof["qr_calls"], "lu_calls": prof.get("lu_calls", 0)}
        if prof4 and prof4.get("families"):
            res["device_ms_by_family_timed_regime"] = {
                "families": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(prof4["families"].items())},
                "contexts": prof4["contexts"], "sweeps_per_s_with_event_records": R * B * a.steps / prof4["dt_profiled"],
                "note": "the %d contexts of the timed region profiled TOGETHER (HIP events on each context's own stream, the same %d steps once more): "
                        "device time summed over the contexts; a family's time includes what its launches waited for kernels of the other "
                        "contexts, so the sum exceeds contexts x wall time of one context alone; qr_apply, lu_update and fact_gemm are parts of decomp"
                        % (prof4["contexts"], a.steps)}
        if cpu is not None:
            res.update(cpu.finish(r0["m"], (r0["m"] + WORKLOAD["s"] - 1) // WORKLOAD["s"]))
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
