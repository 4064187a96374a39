"""Developer tool: how much do the kernels of two families overlap in time?  Reads a rocprofv3 --kernel-trace csv.
    python scripts/overlap_stats.py kernel_trace.csv k_update_decide k_flush"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def iv(key):
    out = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"])
    return out
a, b = iv(sys.argv[2]), iv(sys.argv[3])
ta = sum(e - s for s, e in a); tb = sum(e - s for s, e in b)
j = 0; ov = 0
for s, e in a:
    while j < len(b) and b[j][1] <= s: j += 1
    k = j
    while k < len(b) and b[k][0] < e:
        ov += max(0, min(e, b[k][1]) - max(s, b[k][0])); k += 1
span = max(x[1] for x in a + b) - min(x[0] for x in a + b)
print("%s: %d launches %.1f ms; %s: %d launches %.1f ms; overlap %.1f ms (%.0f %% of the shorter); wall span %.1f ms" % (
    sys.argv[2], len(a), ta / 1e6, sys.argv[3], len(b), tb / 1e6, ov / 1e6, 100.0 * ov / max(1, min(ta, tb)), span / 1e6))
# a few consecutive launches as a time line
ev = sorted([(s, e, "D") for s, e in a] + [(s, e, "F") for s, e in b])
mid = len(ev) // 2
t0 = ev[mid][0]
print("time line (us from an arbitrary origin): " + "  ".join("%s[%d..%d]" % (n, (s - t0) // 1000, (e - t0) // 1000) for s, e, n in ev[mid:mid + 12]))
