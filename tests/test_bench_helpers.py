"""bench.py helpers that can be checked without a GPU: the mapping from the timed scopes of the bench line to the kernels of
the committed rocprofv3 PMC runs (profiles/rNN_pmc_traffic_<workload>.json)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_pmc_traffic_scopes_sum_their_kernels():
    b = _bench()
    d = {k.replace("void ", ""): v for k, v in json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_cfg2.json"))).items()}
    per = lambda v: v["fetch_bytes_x2"] + v["write_bytes"]
    fused = sum(per(v) for k, v in d.items() if k.startswith("k_hit_sub<true,"))
    plain = sum(per(v) for k, v in d.items() if k.startswith("k_hit_sub<false,"))
    assert fused > 0 and plain > 0
    if os.path.exists(os.path.join(ROOT, "profiles", "r02_pmc_traffic_cfg2.json")):
        return  # a newer profile of this workload takes precedence: the arithmetic below is about the r01 file
    assert b.pmc_traffic("k_hit_sub<cut+flt>", "cfg2")[0] == round(fused)   # one launch of each size-class kernel per timed scope
    assert b.pmc_traffic("k_hit_sub", "cfg2")[0] == round(plain)
    assert abs(b.pmc_traffic("k_hit_keys", "cfg2")[0] - 800e6) < 5e6         # 640 MB of strided record reads + 160 MB of keys
    assert b.pmc_traffic("no_such_kernel", "cfg2") == (None, None)
    assert b.pmc_traffic("k_hit_sub", "no_such_workload") == (None, None)   # counters belong to the profiled workload only
