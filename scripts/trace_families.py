"""Kernel-time table of a rocprofv3 kernel trace (csv), launches grouped by kernel name (template arguments kept) and, for the QR / LU
kernels, by grid size:   python scripts/trace_families.py trace.csv [top=40]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    m = re.match(r"(void )?([\w:]+(<[^>(]*>)?)", n)
    key = m.group(2)[:44] if m else n[:44]
    wgs = 0
    if "qr_" in n or "lu_" in n or "zgemm" in n or "trsm" in n:
        wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    tot[(key, wgs)][0] += 1
    tot[(key, wgs)][1] += d
T = sum(v[1] for v in tot.values())
print("total kernel time %.1f ms" % (T / 1e3))
byname = collections.defaultdict(float)
for (k, _), v in tot.items():
    byname[k] += v[1]
print("by kernel:")
for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:20]:
    print("   %-44s %8.1f ms  %5.1f %%" % (k, v / 1e3, 100 * v / T))
print("by kernel and grid:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:top]:
    print("   %-44s wgs %6d  n %5d  avg %8.1f us  total %7.1f ms  %4.1f %%" % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / T))
