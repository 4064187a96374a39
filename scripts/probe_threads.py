"""Developer probe: S kernel contexts of B chains each inside ONE process, swept by S host threads (ctypes releases the GIL
while a sweep runs) -- does one process reach what S worker processes reach?
    python scripts/probe_threads.py S B nsweeps"""
import sys, time, threading, dataclasses, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detqmc_amd import DetSDWBatch, SDWParams
S, B, nsw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p0 = SDWParams(opdim=2, L=16, beta=10.0, s=10, delaySteps=32, stabilisation="qr")
batches = [DetSDWBatch([dataclasses.replace(p0, simindex=g * B + b) for b in range(B)]) for g in range(S)]


def run(batch, n):
    for _ in range(n):
        batch.sweepThermalization()
    batch.kernel_context.synchronize()


def timed(n):
    th = [threading.Thread(target=run, args=(b, n)) for b in batches]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    return time.time() - t0


timed(2)
dt = timed(nsw)
print("threads: S=%d B=%d  %.1f ms/lockstep-sweep  %.1f sweeps/s total (HW queues env %s)" % (S, B, 1e3 * dt / nsw, S * B * nsw / dt, os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
# the same contexts swept one after the other by one thread: group-major issue order
t0 = time.time()
for _ in range(nsw):
    for b in batches:
        b.sweepThermalization()
for b in batches:
    b.kernel_context.synchronize()
dt = time.time() - t0
print("one thread, contexts in turn: %.1f sweeps/s total" % (S * B * nsw / dt), flush=True)
