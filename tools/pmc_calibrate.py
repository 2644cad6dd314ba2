#!/usr/bin/env python3
"""What this GPU sustains for the access patterns the hot path is made of, and what rocprofv3's FETCH_SIZE / WRITE_SIZE report for them.

  python tools/pmc_calibrate.py run [--sizes 800,3200] [--reps 5] [--out FILE]      (on the GPU: times every pattern of csrc/diag.hip with HIP events)
  python tools/pmc_calibrate.py pmc <FETCH_SIZE dir> <WRITE_SIZE dir> <run json>      (joins two `rocprofv3 --pmc X --kernel-trace` passes of the run with it)

MI355X_MICROARCH.md (HBM section) gives one calibration -- FETCH_SIZE reports half the bytes of a wide coalesced streaming read -- and asks for a calibration
"on a known byte count in your own access pattern" for other widths and for WRITE_SIZE.  The patterns here have byte counts known by construction."""
import argparse, collections, csv, ctypes, glob, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

KERNEL_OF = {  # pattern name -> the kernel's name as the trace shows it (prefix)
    "read16_flat": "k_diag_read16_flat", "read16_x8": "void k_diag_read16_x8<0>", "read16_x8+lds_atomics": "void k_diag_read16_x8<1>",
    "read16_x8+lds_atomics_x2": "void k_diag_read16_x8<2>", "read8_of32": "void k_diag_read8_of32<false>", "read8_of32+write8": "void k_diag_read8_of32<true>",
    "write16": "void k_diag_write<16>", "write8": "void k_diag_write<8>", "copy16": "void k_diag_copy<16>", "copy8": "void k_diag_copy<8>",
    "gather32": "void k_diag_gather32<false, 1>", "gather32_ilp4": "void k_diag_gather32<false, 4>", "gather32+cols": "void k_diag_gather32<true, 1>",
    "gather32_ilp4+cols": "void k_diag_gather32<true, 4>", "scatter_runs32": "k_diag_scatter_runs",
    "gather32+cols_win32MB": None, "gather32+cols_win64MB": None, "gather32+cols_win128MB": None, "gather32+cols_win256MB": None, "gather32+cols_win512MB": None,  # (same kernel as gather32+cols: timing only)
    "read16_x8+lds_atomics+rows_out": "void k_diag_read16_x8<1, true>",
}


def run(args):
    import miniasm_amd
    L = miniasm_amd.lib()
    L.mahip_create.restype = ctypes.c_void_p
    L.mahip_create.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.mahip_diag_name.restype = ctypes.c_char_p
    L.mahip_diag_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.mahip_strerror.restype = ctypes.c_char_p
    c = L.mahip_create(0, None)
    if not c:
        sys.exit("mahip_create failed: %s" % L.mahip_strerror().decode())
    rows = []
    for mb in [int(x) for x in args.sizes.split(",")]:
        for p in range(L.mahip_diag_patterns()):
            if args.patterns and L.mahip_diag_name(p).decode() not in args.patterns.split(","):
                continue
            ms, mv = ctypes.c_double(), ctypes.c_double()
            if L.mahip_diag_run(c, p, mb << 20, args.reps, ctypes.byref(ms), ctypes.byref(mv)) != 0:
                sys.exit("pattern %d: %s" % (p, L.mahip_strerror().decode()))
            name = L.mahip_diag_name(p).decode()
            rows.append({"pattern": name, "source_MB": mb, "moved_bytes": mv.value, "best_ms": ms.value, "TB_per_s": mv.value / ms.value / 1e9})
            print("%-26s %5d MB  moved %7.3f GB  %8.3f ms  %6.2f TB/s" % (name, mb, mv.value / 1e9, ms.value, mv.value / ms.value / 1e9), flush=True)
    L.mahip_destroy(ctypes.c_void_p(c))
    if args.out:
        json.dump({"what": "csrc/diag.hip patterns, fastest of %d launches each (HIP events)" % args.reps, "rows": rows}, open(args.out, "w"), indent=1)


def counters(d, want):
    acc = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == want:
                    acc[row.get("Kernel_Name", "?").split("(")[0]].append(float(row.get("Counter_Value", 0)) * 1024)
    return acc


def pmc(args):
    fe, wr = counters(args.fetch_dir, "FETCH_SIZE"), counters(args.write_dir, "WRITE_SIZE")
    rows = json.load(open(args.run_json))["rows"]
    sizes = sorted({r["source_MB"] for r in rows})
    out = []
    for r in rows:
        k = KERNEL_OF[r["pattern"]]
        if k is None:
            continue
        # the run launches every pattern `reps` times per size, sizes in ascending order: the launches of a kernel split evenly over the sizes
        def pick(acc):
            v = acc.get(k, [])
            per = len(v) // len(sizes) if sizes else 0
            i = sizes.index(r["source_MB"])
            part = v[i * per:(i + 1) * per]
            return sum(part) / len(part) if part else None
        f, w = pick(fe), pick(wr)
        mb = r["source_MB"] << 20
        n_g = 1 << ((mb // 32).bit_length() - 1)
        known_r = {"write16": 0, "write8": 0, "gather32": 32 * n_g, "gather32_ilp4": 32 * n_g, "gather32+cols": 32 * n_g, "gather32_ilp4+cols": 32 * n_g}.get(r["pattern"], mb)
        known_w = r["moved_bytes"] - known_r
        row = dict(r, known_read_bytes=known_r, known_write_bytes=known_w, FETCH_SIZE_bytes=f, WRITE_SIZE_bytes=w,
                   fetch_over_known=(f / known_r if f is not None and known_r else None), write_over_known=(w / known_w if w is not None and known_w > (1 << 20) else None))
        out.append(row)
        print("%-26s %5d MB  read known %6.2f GB counter %s (x%s)   write known %6.2f GB counter %s (x%s)" % (
            r["pattern"], r["source_MB"], known_r / 1e9, "%6.2f GB" % (f / 1e9) if f is not None else "-", "%.2f" % row["fetch_over_known"] if row["fetch_over_known"] else "-",
            known_w / 1e9, "%6.2f GB" % (w / 1e9) if w is not None else "-", "%.2f" % row["write_over_known"] if row["write_over_known"] else "-"))
    json.dump({"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) of `tools/pmc_calibrate.py run`, per launch, against the byte "
                       "counts the patterns have by construction (a 32-byte random fetch is counted as 32 B: anything above is the memory system's granularity)",
               "rows": out}, sys.stdout if not args.out else open(args.out, "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("run"); a.add_argument("--sizes", default="800,3200"); a.add_argument("--reps", type=int, default=5); a.add_argument("--out")
    a.add_argument("--patterns", default="", help="comma list of pattern names (default: all; the PMC passes leave the windowed gathers out -- they share a kernel)")
    b = sub.add_parser("pmc"); b.add_argument("fetch_dir"); b.add_argument("write_dir"); b.add_argument("run_json"); b.add_argument("--out")
    args = ap.parse_args()
    run(args) if args.cmd == "run" else pmc(args)
