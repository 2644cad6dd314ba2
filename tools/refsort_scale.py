#!/usr/bin/env python3
"""host/refsort.c on the keys of one pafgen input with 8 .. 128 threads (CPU alone): does the part below the top level scale?
usage: tools/refsort_scale.py [--reads R --lines N --seed S] [--threads 8,16,32,64,128]"""
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
    ap.add_argument("--reads", type=int, default=2000000)
    ap.add_argument("--lines", type=int, default=100000000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--threads", default="8,16,32,64,128")
    ap.add_argument("--workdir", default=os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"))
    a = ap.parse_args()
    os.environ["MA_REFSORT_TIMING"] = "1"
    import miniasm_amd as ma
    import refapi as R
    os.makedirs(a.workdir, exist_ok=True)
    paf = os.path.join(a.workdir, "rs_scale_r%d_n%d_s%d.paf" % (a.reads, a.lines, a.seed))
    if not os.path.exists(paf):
        R.pafgen(paf, a.reads, a.lines, a.seed, ["-L", "uniform", "-d", "0.35", "-x", "0.03"])
    L = ma.lib()
    L.ma_set_log_path(b"/dev/null")
    ing = ma.Ingest(paf, ma.default_opt())
    keys = np.ascontiguousarray(ing.hits["qns"])
    ing.free_hits()
    n = len(keys)
    L.ma_refsort_perm.restype = C.c_int
    L.ma_refsort_perm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    perm = np.zeros(n, dtype=np.uint32)
    for t in a.threads.split(","):
        os.environ["MA_THREADS"] = t
        L.setenv_c = None
        C.CDLL(None).setenv(b"MA_THREADS", t.encode(), 1)
        t0 = time.perf_counter()
        assert L.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        print("## %s threads: %d keys, %.3f s" % (t, n, time.perf_counter() - t0), flush=True)
    ing.close()


if __name__ == "__main__":
    main()
