"""Sum rocprofv3 --pmc counters per kernel: python scripts/pmc_summary.py counter_collection.csv substring"""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-32s %.4g  (%d dispatches)" % (k, acc[k], n[k]))
