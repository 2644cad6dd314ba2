#!/usr/bin/env python3
"""one screen of a rocprofv3 *kernel_stats.csv: name, calls, average, total"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.3f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:n]:
    print("%-56s x%-5s avg %9.1f us  total %8.3f ms" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
