"""Per-round durations of the update kernels in a rocprofv3 kernel trace (csv) of a batched sweep: the k-th launch of a kernel
within a slice is round k % rounds.   python scripts/trace_rounds.py trace.csv [rounds=8]"""
import csv, sys, collections
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for name in ("k_update_decide", "k_update_gather", "k_flush"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    n = len(d) // rounds * rounds
    d = d[len(d) - n:]                                   # the last complete slices
    per = collections.defaultdict(list)
    for i, x in enumerate(d):
        per[i % rounds].append(x)
    print("%s: %d launches, total %.1f ms" % (name, n, sum(d) / 1e3))
    for k in range(rounds):
        v = sorted(per[k])
        print("   round %d: mean %7.1f us  median %7.1f  min %7.1f  max %7.1f   share %.1f %%" % (
            k, sum(v) / len(v), v[len(v) // 2], v[0], v[-1], 100 * sum(v) / sum(d)))
