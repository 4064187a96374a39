"""Average launch durations: bench.py's live HIP-event averages (JSON line) next to the rocprofv3 --kernel-trace --stats summary of
the same command.   python scripts/compare_events_vs_rocprof.py bench.json kernel_stats.csv
The csv covers every launch of the process (set-up, warm-up, timed steps, profiled steps), the JSON line the profiled steps only:
with enough warm-up for the acceptance to settle the two agree to a few per cent."""
import csv, json, sys
d = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
rows = list(csv.DictReader(open(sys.argv[2])))


def stat(*keys):
    calls, tot = 0, 0.0
    for r in rows:
        if any(k in r["Name"] for k in keys):
            calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])
    return calls, (tot / calls / 1e3 if calls else 0.0)


names = {"flush": ("k_flush<true, true, 1, 0", "k_flush<true, false, 1, 0", "k_flush<false"), "gemm": ("k_zgemm<2, 2, true, 0", "k_zgemm<2, 2, false, 0", "k_zgemm<1, 1, true, 0", "k_zgemm<1, 1, false, 0"),
         "bmult": ("k_bmult_chain",), "qr_apply": ("k_qr_apply",), "gather": ("k_update_gather",), "decide": ("k_update_decide",)}
print("%-10s %22s %26s" % ("family", "HIP events (JSON line)", "rocprofv3 --stats (all launches)"))
for e in [d["roofline"]] + d["roofline_other_kernels"]:
    if e["family"] not in names:
        continue
    c, avg = stat(*names[e["family"]])
    print("%-10s %9.1f us x %6d %12.1f us x %6d   ratio %.3f" % (e["family"], e["avg_launch_us"], e["launches"], avg, c, e["avg_launch_us"] / avg if avg else 0))
