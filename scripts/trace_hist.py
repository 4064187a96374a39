"""Duration histogram of one kernel in a rocprofv3 kernel trace (csv): python scripts/trace_hist.py trace.csv k_zgemm"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
print(len(d), "launches; min %.0f median %.0f max %.0f us" % (d[0], d[len(d) // 2], d[-1]))
h = collections.Counter(int(x // 100) * 100 for x in d)
for k in sorted(h): print("%6d-%6d us: %d" % (k, k + 100, h[k]))
