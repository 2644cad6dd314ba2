#!/usr/bin/env python3
"""Times host/refsort.c (the walk that reproduces the reference's unstable hit sort) on the keys of a pafgen input, on the CPU alone, and -- with
--check, when oracle/_ref is built -- compares the permutation with what the reference's radix_sort_hit does to the same records.
usage: tools/refsort_time.py [--reads R --lines N --seed S] [--check] [--threads T] [-- pafgen options, default: -L uniform -d 0.35 -x 0.03]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--lines", type=int, default=10000000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--threads", default=None)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--workdir", default=os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"))
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    if a.threads:
        os.environ["MA_THREADS"] = a.threads
    os.environ["MA_REFSORT_TIMING"] = "1"
    import miniasm_amd as ma
    import refapi as R
    extra = a.extra or ["-L", "uniform", "-d", "0.35", "-x", "0.03"]
    os.makedirs(a.workdir, exist_ok=True)
    paf = os.path.join(a.workdir, "rs_r%d_n%d_s%d_%s.paf" % (a.reads, a.lines, a.seed, "".join(extra).replace("-", "").replace(".", "")))
    if not os.path.exists(paf):
        R.pafgen(paf, a.reads, a.lines, a.seed, extra)
    L = ma.lib()
    L.ma_set_log_path(b"/dev/null")
    ing = ma.Ingest(paf, ma.default_opt())
    keys = np.ascontiguousarray(ing.hits["qns"])
    n = len(keys)
    L.ma_refsort_perm.restype = C.c_int
    L.ma_refsort_perm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    perm = np.zeros(n, dtype=np.uint32)
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        assert L.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        print("ma_refsort_perm: %d keys, %.3f s" % (n, time.perf_counter() - t0), flush=True)
    if a.check:
        LR = R.ref()
        LR.radix_sort_hit.argtypes = [C.c_void_p, C.c_void_p]
        LR.radix_sort_hit.restype = None
        hits = np.zeros(n, dtype=ma.HIT_DT)
        hits["qns"] = keys
        hits["tn"] = np.arange(n, dtype=np.uint32)
        t0 = time.perf_counter()
        LR.radix_sort_hit(hits.ctypes.data, hits.ctypes.data + n * 32)
        print("reference radix_sort_hit: %.3f s" % (time.perf_counter() - t0))
        same = bool((hits["tn"] == perm).all())
        print("same order as the reference: %s" % same)
        if not same:
            sys.exit(1)
    ing.close()


if __name__ == "__main__":
    main()
