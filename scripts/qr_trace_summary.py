"""Summarise a rocprofv3 kernel trace (csv) of scripts/probe_qr.py: time by kernel and grid size."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[key][0] += 1; acc[key][1] += d
tot = sum(v[1] for v in acc.values())
byname = collections.defaultdict(float)
for (nm, g), v in acc.items(): byname[nm] += v[1]
for nm, t in sorted(byname.items(), key=lambda x: -x[1])[:8]:
    print("%-42s %8.1f us total (%.1f%%)" % (nm, t, 100 * t / tot))
    for (n2, g), v in sorted(acc.items()):
        if n2 == nm: print("      grid.x %4d  calls %4d  avg %8.1f us  sum %9.1f" % (g, v[0], v[1] / v[0], v[1]))
