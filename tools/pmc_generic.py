#!/usr/bin/env python3
"""Mean value per launch of every counter found in rocprofv3 --pmc counter_collection CSVs, per kernel.
usage: pmc_generic.py <dir> [<dir> ...] [--filter regex] [--each]  -> table on stdout (--each: every launch in dispatch order)"""
import csv, glob, os, re, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    dirs = [d for d in dirs if d != flt]
each = "--each" in sys.argv
rows = []
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "?").split("(")[0]
                if flt and not re.search(flt, name):
                    continue
                if each:
                    rows.append((int(row.get("Dispatch_Id", 0) or 0), name, row.get("Counter_Name"), float(row.get("Counter_Value", 0))))
                a = acc[name][row.get("Counter_Name")]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n, s = acc[k][c]
        print("   %-28s launches %4d  mean %16.1f" % (c, n, s / max(n, 1)))
if each:
    for did, name, c, v in sorted(rows):
        print("dispatch %4d  %-40s %-14s %16.1f" % (did, name[:40], c, v))
