"""Rows of a rocprofv3 --stats kernel_stats csv whose kernel name contains one of the given substrings:
python scripts/stats_pick.py kernel_stats.csv k_bmult k_flush"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(s in r["Name"] for s in sys.argv[2:]):
        print("%-58s calls %6s  avg %9.1f us  min %9.1f  max %9.1f  total %8.1f ms" % (
            r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
