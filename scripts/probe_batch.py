"""Developer probe: throughput of B batched replicas (one context, grid.z = chain) vs B."""
import sys, time, dataclasses
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detqmc_amd import DetSDWBatch, SDWParams

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
beta = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stab = sys.argv[4] if len(sys.argv) > 4 else "qr"
Bs = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1, 4, 8, 16]
prof = len(sys.argv) > 6 and sys.argv[6] == "prof"
p0 = SDWParams(opdim=int(os.environ.get("DQMC_OPDIM", "2")), L=L, beta=beta, s=10, delaySteps=int(os.environ.get("DQMC_DELAY_STEPS", "16")), stabilisation=stab)
for B in Bs:
    t0 = time.time()
    batch = DetSDWBatch([dataclasses.replace(p0, simindex=b, r=p0.r + 0.01 * b) for b in range(B)], sub_batches=int(os.environ.get("DQMC_SUB_BATCHES", "1")))
    tinit = time.time() - t0
    ctx = batch.kernel_context
    batch.sweepThermalization(); batch.sweepThermalization()
    ctx.synchronize()
    if prof:
        ctx.profile_enable(True)
    t0 = time.time()
    for i in range(nsw):
        batch.sweepThermalization()
    ctx.synchronize()
    dt = time.time() - t0
    print("B=%d init %.2fs  %.1f ms/lockstep-sweep  %.2f sweeps/s total" % (B, tinit, 1e3 * dt / nsw, B * nsw / dt), flush=True)
    if prof:
        pr = ctx.profile_read()
        print("   ", {k: (round(v[0] / nsw, 1), v[1] // nsw) for k, v in pr.items() if isinstance(v, tuple)}, flush=True)
    # counters since create (profiling was never switched on in a PMC run): work of the update kernels in THIS run
    pr = ctx.profile_read()
    print("COUNTERS blocks_nonempty=%d updates_accepted=%d qr_calls=%d chains=%d sweeps_total=%d n_g=%d" % (
        pr["blocks_nonempty"], pr["updates_accepted"], pr["qr_calls"], pr["chains"], nsw + 2, batch.chain(0).info.n_g), flush=True)
    if int(os.environ.get("DQMC_DECIDE_TIMING", "0")):
        ctx.update_state()          # prints the decision kernel's phase timers (library built with -DDQMC_DECIDE_TIMING)
    batch.close()
