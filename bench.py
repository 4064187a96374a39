#!/usr/bin/env python3
"""Benchmark harness: DQMC sweeps/sec for the BASELINE.json headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One process per GPU (for N > 1 launched by torch.distributed.run).  A "step" is one
sweepThermalization() of every replica the rank drives: --workers W worker processes per GPU (own HIP
runtime and hardware queues each), each holding --batch B independent chains that advance in lockstep
through ONE kernel context (every launch carries all B chains, grid.z = chain).  value = all sweeps of all
chains / time; `sweeps_per_s_per_chain` is the single-chain rate.  Workload: SDW O(2), L=16, beta=10,
dtau=0.1 (m=100), s=10, checkerboard, delayed updates (delaySteps=16), no fermion measurements (SURVEY.md
section 8d).  Every chain is an independent Markov chain with its own RNG stream (simindex), like the
replicas the reference's DetQMC / DetQMCPT run one per MPI process; there is no
data-path collective, so the value is the sum over ranks and scaling is weak.  Rank 0 prints ONE
JSON line with `roofline` (dominant kernel, timed live with HIP events on the kernel's own stream)
and, at N=1, `cpu_baseline` (the real reference binary from oracle/_ref when it runs on this host,
else the numpy oracle port) on a bounded sample of the same workload.
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
WORKLOAD = dict(opdim=2, L=16, beta=10.0, dtau=0.1, s=10, delaySteps=int(os.environ.get("DQMC_DELAY_STEPS", "32")), r=-1.0, c=3.0, u=1.0, lambda_=1.0,
                mu=-0.5, txhor=-1.0, txver=-0.5, tyhor=0.5, tyver=1.0, bc="pbc", accRatio=0.5,
                rngSeed=1020304050,
                # same Green's functions and Markov chain as the reference-exact "svd" mode (tests), ~10x cheaper
                stabilisation=os.environ.get("DQMC_STABILISATION", "qr"))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F64_PEAK_TF = 78.6        # CDNA4 v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s (= fp64 vector rate)
DEFAULT_BATCH = 128            # chains per kernel context (lockstep batch); 4 x 128 chains = 123 GB of the 288 GB HBM
DEFAULT_WORKERS = 4            # contexts per GPU: the latency-bound kernels of one overlap the streaming kernels of the others


def cpu_baseline(max_seconds=200):
    """Reference CPU sweeps/s on this host for the same workload (bounded sample)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_fast_o2")
    args = ["L=16", "beta=10", "dtau=0.1", "s=10", "delaySteps=16", "opdim=2", "mode=time", "warmup=0", "sweeps=1"]
    if os.path.exists(exe):
        try:
            env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
            out = subprocess.run([exe, "/tmp"] + args, capture_output=True, text=True, timeout=max_seconds, env=env)
            for line in out.stdout.splitlines():
                if line.startswith("REF_TIMING"):
                    kv = dict(tok.split("=") for tok in line.split()[1:])
                    return {"value": float(kv["sweeps_per_s"]), "unit": "sweeps/s", "cores": 1, "kind": "reference",
                            "sample": "crstnbr/detqmc DetSDW<CB_ASSAAD_BERG,2> built from the reference sources "
                                      "(-O3 -ffast-math -mavx2 -mfma, MKL 1 thread): 1 sweepThermalization() after "
                                      "init, %s s" % kv["seconds"]}
        except Exception as e:                      # binary cannot run on this host: use the port
            sys.stderr.write("reference binary unusable (%r), timing the oracle port instead\n" % (e,))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))   # cpu_baseline leg: the oracle is what is timed here
    from detsdw_oracle import DetSDWOracle, SDWParams as OP
    kw = {k: v for k, v in WORKLOAD.items() if k in OP.__dataclass_fields__ and k != "stabilisation"}
    o = DetSDWOracle(OP(**kw))
    # bounded sample: the first slices of a down sweep (local updates + wrap) for ~20 s, one stabilisation step,
    # extrapolated to the m slices and n stabilisation steps of a full sweep
    m, n, s_ = o.m, o.n, o.s
    t0 = time.time()
    k, nsl = m, 0
    while nsl < (m - (n - 1) * s_) and (time.time() - t0 < 20.0 or nsl == 0):
        o.updateInSliceThermalization(k)
        o.wrapDownGreen(k)
        k -= 1
        nsl += 1
    t_slice = (time.time() - t0) / nsl
    sweep_s = m * t_slice
    sample = "%d of %d time slices (local updates + wrap) in %.1f s" % (nsl, m, time.time() - t0)
    if k == (n - 1) * s_:                      # the slices of the top interval are done: time one advanceDownGreen too
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


def worker(a, readline=None, emit=None):
    """One kernel context with a.batch chains in lockstep: build, warm up, wait for GO, run, report."""
    import dataclasses
    readline = readline or sys.stdin.readline
    emit = emit or (lambda line: print(line, flush=True))
    from detqmc_amd import DetSDWBatch, SDWParams
    B = max(1, a.batch)
    p0 = SDWParams(device=a.device, **WORKLOAD)
    batch = DetSDWBatch([dataclasses.replace(p0, simindex=a.simindex * B + i) for i in range(B)])
    ctx = batch.kernel_context
    for _ in range(a.warmup):
        batch.sweepThermalization()
    blocks0 = 0
    if a.profile:
        blocks0 = ctx.profile_read()["blocks_nonempty"]
        ctx.profile_enable(True)
    ctx.synchronize()
    emit("READY")
    if readline().strip() != "GO":
        return
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batch.sweepThermalization()          # one C call per lockstep sweep of all B chains
    ctx.synchronize()
    dt = time.perf_counter() - t0
    info = batch.chain(0).info
    out = {"dt": dt, "n_g": info.n_g, "m": info.m,
           "acceptance": [batch.chain(i).info.lastAccRatioLocal_phi for i in range(min(B, 4))]}
    if a.profile:
        prof = ctx.profile_read()
        out["prof"] = {k: list(v) if isinstance(v, tuple) else v for k, v in prof.items()}
        out["prof"]["blocks_nonempty"] -= blocks0
    emit("RESULT " + json.dumps(out))
    # worker 0 is then asked to repeat the steps ALONE on the GPU: kernel durations free of the other contexts
    if readline().strip() == "SOLO":
        blocks0 = ctx.profile_read()["blocks_nonempty"]
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            batch.sweepThermalization()
        ctx.synchronize()
        prof = ctx.profile_read()
        prof["blocks_nonempty"] -= blocks0
        emit("SOLO " + json.dumps({"dt": time.perf_counter() - t0,
                                   "prof": {k: list(v) if isinstance(v, tuple) else v for k, v in prof.items()}}))
    batch.close()


class InprocWorker:
    """The worker protocol inside this process (one context, no child processes): what `rocprofv3 -- python3 bench.py
    --inprocess` can profile -- on this pool a profiled process must not spawn GPU children."""

    def __init__(self, a):
        import copy
        import queue
        import threading
        self.inq, self.outq = queue.Queue(), queue.Queue()
        wa = copy.copy(a)
        wa.profile = True
        self.stdin, self.stdout = self, self
        self.th = threading.Thread(target=worker, args=(wa, self.inq.get, self.outq.put), daemon=True)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("DQMC_WORKERS_PER_GPU", str(DEFAULT_WORKERS))),
                    help="worker processes (kernel contexts) per GPU")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DQMC_CHAINS_PER_CONTEXT", str(DEFAULT_BATCH))),
                    help="independent Markov chains per context, advanced in lockstep (grid.z = chain)")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE context in this process instead of worker processes (for rocprofv3)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--simindex", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--profile", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.worker:
        return worker(a)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    dist = None
    torch = None
    # DQMC_BENCH_BACKEND=gloo + DQMC_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a one-GPU box
    # (all ranks drive device 0, the ranks themselves stay off the GPU); the real runs use nccl (= RCCL)
    backend = os.environ.get("DQMC_BENCH_BACKEND", "nccl")
    if os.environ.get("DQMC_BENCH_ONE_DEVICE"):
        local = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

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
    procs = []
    if a.inprocess:
        a.device, a.simindex = local, rank
        procs.append(InprocWorker(a))
    for i in range(0 if a.inprocess else R):
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--device", str(local), "--simindex",
               str(rank * R + i), "--steps", str(a.steps), "--warmup", str(a.warmup), "--batch", str(B)] + (["--profile"] if i == 0 else [])
        procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env))

    def read_tag(p, tag):
        while True:
            line = p.stdout.readline()
            if not line:
                raise SystemExit("bench worker died (exit code %s)" % p.poll())
            if line.startswith(tag):
                return line[len(tag):].strip()

    for p in procs:
        read_tag(p, "READY")            # replicas built, warm-up sweeps done, devices idle
    fence()
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write("GO\n")
        p.stdin.flush()
    results = [json.loads(read_tag(p, "RESULT")) for p in procs]    # each worker synchronised its stream
    fence()
    dt = time.perf_counter() - t0
    solo = None
    for i, p in enumerate(procs):
        p.stdin.write("SOLO\n" if (i == 0 and rank == 0) else "QUIT\n")
        p.stdin.flush()
    if rank == 0:
        solo = json.loads(read_tag(procs[0], "SOLO"))
    for p in procs:
        p.wait()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        r0 = results[0]
        n = r0["n_g"]
        N, MSF, D, OPD = n // 2, 2, WORKLOAD["delaySteps"], WORKLOAD["opdim"]

        def rooflines(rawprof, sharing):
            prof = {k: (tuple(v) if isinstance(v, list) else v) for k, v in rawprof.items()}

            def hbm(name, kernel, ms, launches, bytes_per_chain, note, work_launches=None):
                """achieved = algorithmic bytes of the launches that had work / device time of ALL launches of the kernel;
                avg_launch_us is over all launches, like rocprofv3's AverageNs."""
                work = launches if work_launches is None else work_launches
                bytes_per_launch = bytes_per_chain * B          # every launch carries all B chains of the context
                us = 1e3 * ms / max(launches, 1)
                ach = bytes_per_launch * work / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                return {"family": name, "kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": bytes_per_launch,
                        "avg_launch_us": us, "launches": launches, "launches_with_work": work, "device_ms": ms,
                        "chains_per_launch": B, "contexts_sharing_the_gpu": sharing, "note": note}

            roofs = []
            # decision kernel: per proposal (OPDIM+1) uniforms, 2*OPDIM field values of the neighbouring slices, cosh/sinh,
            # G[c,I], G[I,c], G[c,c], G[c,prev], G[prev,c] with |I| = MSF*D/2 on average; N proposals per slice over
            # the launches that found work
            cand_bytes = (OPD + 1) * 8 + 2 * OPD * 8 + 16 + (2 * MSF * (MSF * D // 2) + 3 * MSF * MSF) * 16
            nblocks = max(prof["blocks_nonempty"], 1)
            nslices = prof["decide"][1] // ((N + D - 1) // D)
            roofs.append(hbm("decide", "k_update_decide<2>", *prof["decide"], N * cand_bytes * nslices / nblocks,
                             "sequential Metropolis chain of one slice: ONE workgroup per chain by construction -- a latency-bound "
                             "kernel, HBM is not its limit; blocks without work exit at once", nblocks))
            if WORKLOAD["stabilisation"] == "svd":
                r = hbm("decomp", "k_jacobi_round<8,2>", prof["decomp_round_ms"], prof["decomp_rounds"], 4.0 * n * n * 16.0,
                        "one Jacobi round reads and writes every column of A and V once")
                roofs.append(r)
            else:
                # k_qr_apply_reg reads and writes the columns it is given once per launch; the launch sequence of
                # run_qr (kernels_qr.hip): groups of 4 panels, look-ahead launches for the group's own columns
                def qr_apply_traffic(nn):
                    tot, p, npan = 0.0, 0, (nn + 15) // 16
                    while p < npan:
                        j0 = p * 16
                        left = nn - j0
                        if nn <= 512 and left > 64:
                            for rows, ncols in ((left, 16), (left, 32), (left - 32, 16), (left, left - 64)):
                                tot += 2 * 16.0 * rows * ncols
                            p += 4
                        elif nn <= 512 and left > 32:
                            tot += 2 * 16.0 * left * 16 + 2 * 16.0 * left * (left - 32)
                            p += 2
                        else:
                            nbp = min(16, left)
                            if left - nbp > 0:
                                tot += 2 * 16.0 * left * (left - nbp)
                            p += 1
                    p = npan - 1
                    while p >= 0:                       # formation of Q, reflectors in reverse order
                        cnt = 1
                        if nn <= 512 and min(16, nn - p * 16) == 16:
                            cnt = 4 if p >= 3 else (2 if p >= 1 else 1)
                        lo = p - (cnt - 1)
                        tot += 2 * 16.0 * (nn - lo * 16) ** 2
                        p -= cnt
                    return tot
                qr_bytes = qr_apply_traffic(n)
                calls = max(prof["qr_calls"], 1)
                napply = max(prof["decomp_rounds"], 1)
                roofs.append(hbm("qr_apply", "k_qr_apply_reg", prof["decomp_round_ms"], napply, qr_bytes * calls / napply,
                                 "block reflectors of up to 4 panels applied to the trailing matrix / to Q while the columns stay in "
                                 "registers: one read + one write per launch; bytes = average over the launches of a factorisation"))
                rest_ms = max(prof["decomp"][0] - prof["decomp_round_ms"], 0.0)
                rest_l = max(prof["decomp"][1] - napply, 1)
                roofs.append(hbm("qr_rest", "k_qr_panel, triangular solve, pivoting glue", rest_ms, rest_l, 0.0,
                                 "panel factorisations (one workgroup per chain, a chain of dependent reductions: latency bound) and the "
                                 "small kernels around the QR; no roofline claimed"))
            roofs.append(hbm("bmult", "k_bmult_chain", *prof["bmult"], 2 * 16.0 * n * n, "one read + one write of A per chain of slices"))
            roofs.append(hbm("gather", "k_update_gather", *prof["gather"], 4 * 16.0 * n * MSF * D,
                             "X = G[:,I] W and Gr = G[I,:] - E; blocks without work exit at once", nblocks))
            roofs.append(hbm("flush", "k_flush (G += X Gr)", *prof["flush"], 2 * 16.0 * n * n,
                             "read-modify-write of G once per delayed-update block that accepted an update; the launches of the "
                             "other blocks exit at once", nblocks))
            # HBM bytes of k_flush from the PMC counters (collected separately, profiles/r01_pmc_flush_*.json: FETCH_SIZE and
            # WRITE_SIZE passes, FETCH doubled as MI355X_MICROARCH.md prescribes), per launch that had work, if the file
            # was taken with this run's chains per launch and delaySteps
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_flush_b%d_d%d.json" % (B, D))
            if os.path.exists(pmc):
                try:
                    for r_ in roofs:
                        if r_["family"] == "flush":
                            r_["traffic"] = json.load(open(pmc))["hbm_bytes_per_launch_with_work"]
                            r_["traffic_note"] = "per launch with work, rocprofv3 --pmc, " + os.path.basename(pmc)
                except Exception:
                    pass
            gms, gl = prof["gemm"]
            tf = prof["gemm_flops"] / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
            roofs.append({"family": "gemm", "kernel": "k_zgemm<2,2>", "bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TF,
                          "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TF, "traffic": None,
                          "algorithmic_flops_per_launch": prof["gemm_flops"] / max(gl, 1), "avg_launch_us": 1e3 * gms / max(gl, 1),
                          "launches": gl, "launches_with_work": gl, "device_ms": gms, "chains_per_launch": B, "contexts_sharing_the_gpu": sharing,
                          "note": "n_g^3 complex products on v_mfma_f64_16x16x4_f64"})
            roofs.sort(key=lambda r: -r["device_ms"])
            return prof, roofs

        prof, roofs = rooflines(r0["prof"], R)
        fam = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items() if isinstance(v, tuple) and k != "jacobi"}
        _, roofs_solo = rooflines(solo["prof"], 1)
        solo_by_family = {r["family"]: r for r in roofs_solo}
        res = {
            "metric": "DQMC sweeps/sec (SDW-O2 L=16 beta=10 fp64)",
            "value": world * R * B * a.steps / dt,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "sweeps_per_s_per_chain": a.steps / dt,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (random initial field, fixed seed)",
            "config": {"workload": "DetSDW O(2) L=16 beta=10 dtau=0.1 s=10 checkerboard delayed(%d) " % WORKLOAD["delaySteps"] +
                                   "sweepThermalization, %d independent chains per GPU (%d kernel contexts x %d lockstep chains), stabilisation=%s"
                                   % (R * B, R, B, WORKLOAD["stabilisation"]),
                       "n_g": n, "m": r0["m"], "replicas_per_gpu": R * B, "contexts_per_gpu": R,
                       "chains_per_context": B},
            # timed region, context 0 while all contexts share the GPU: durations include waiting for the others
            "roofline": roofs[0],
            "roofline_other_kernels": roofs[1:],
            # the same kernels right after the timed region with ONE context alone on the GPU (clean durations;
            # this is what the rocprofv3 summary under profiles/ shows)
            "roofline_solo_context": solo_by_family[roofs[0]["family"]],
            "roofline_solo_context_other_kernels": [r for r in roofs_solo if r["family"] != roofs[0]["family"]],
            "solo_context_sweeps_per_s": B * a.steps / solo["dt"],
            "device_ms_by_family_context0": fam,
            "decompositions_context0": {"svd_calls": prof["svd_calls"], "jacobi_sweeps": prof["svd_sweeps_total"],
                                      "max_sweeps": prof["svd_sweeps_max"], "qr_calls": prof["qr_calls"]},
            "acceptance": r0["acceptance"],
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
