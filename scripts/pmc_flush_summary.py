"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) of scripts/probe_batch.py into the
profiles/r01_pmc_flush_b<B>_d<D>.json that bench.py reads for roofline.traffic.
    python scripts/pmc_flush_summary.py fetch.csv write.csv B D out.json"""
import csv, json, sys
fetch_csv, write_csv, B, D, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]


def collect(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_flush" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    work = [v for v in vals if v > 1024.0]                   # (KiB) launches of blocks without an accepted update return at once
    return dict(dispatches=len(vals), with_work=len(work), KiB_avg_with_work=sum(work) / max(1, len(work)), KiB_avg_all=sum(vals) / max(1, len(vals)))


f, w = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
read_b = 2.0 * f["KiB_avg_with_work"] * 1024.0              # gfx950: FETCH_SIZE counts half of wide coalesced reads
write_b = w["KiB_avg_with_work"] * 1024.0
n = 512
doc = {
    "kernel": "k_flush", "chains_per_launch": B, "delaySteps": D,
    "workload": "DetSDW O(2) L=16 beta=10 (n_g=512), delaySteps %d, %d chains per launch (bench.py's context), DQMC_DELAY_STEPS=%d "
                "scripts/probe_batch.py 16 10 1 qr %d, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes" % (D, B, D, B),
    "counters": {"FETCH_SIZE": f, "WRITE_SIZE": w},
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced 16 B/lane reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
    "hbm_bytes_per_launch_with_work": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
    "algorithmic_bytes_per_launch": 2 * 16 * n * n * B,
    "note": "launches of blocks without accepted updates return at once (counter ~ 0) and are excluded from the averages; chains of a launch that accepted nothing in the block neither read nor write G, so the average can stay below the all-chains algorithmic figure",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("hbm_bytes_per_launch_with_work", "algorithmic_bytes_per_launch")}))
