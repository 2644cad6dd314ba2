#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) per launch.
usage: pmc_summary.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run>   -> JSON on stdout
gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request, i.e. HALF the bytes of a stream; `fetch_bytes_x2` applies that correction,
`fetch_bytes_raw` is the counter as is.  Calibrated in round 4 against byte counts known by construction (tools/pmc_calibrate.py, profiles/r04_pmc_calibration.json):
the factor 2 holds for every streaming pattern of the hot path (16 and 8 bytes per lane, 8 bytes of every 32-byte record), a random 32-byte fetch shows as 64 B
(x2 = the 128-byte line it moves), WRITE_SIZE is exact for all of them."""
import csv, glob, json, os, sys, collections

def load(d, want):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != want:
                    continue
                name = row.get("Kernel_Name", "?").split("(")[0]
                a = acc[name]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0))
    return {k: (v[0], v[1] / max(v[0], 1)) for k, v in acc.items()}

fe = load(sys.argv[1], "FETCH_SIZE")
wr = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr)):
    nf, f = fe.get(k, (0, 0.0))
    nw, w = wr.get(k, (0, 0.0))
    out[k] = {"launches": max(nf, nw), "fetch_bytes_raw": f * 1024, "fetch_bytes_x2": 2 * f * 1024, "write_bytes": w * 1024}
print(json.dumps(out, indent=1))
print("kernels:", len(out), file=sys.stderr)
