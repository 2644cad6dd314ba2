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
    if any(os.path.exists(os.path.join(ROOT, "profiles", "%s_pmc_traffic_cfg2.json" % r)) for r in ("r02", "r03a", "r03")):
        return  # a newer profile of this workload takes precedence: the arithmetic below is about the r01 file
    names, src = b.pmc_profile("cfg2")
    assert src.startswith("profiles/r01_pmc_traffic_cfg2.json")
    assert b.pmc_traffic(names, "k_hit_sub<cut+flt>") == round(fused)   # one launch of each size-class kernel per timed scope
    assert b.pmc_traffic(names, "k_hit_sub") == round(plain)
    assert abs(b.pmc_traffic(names, "k_hit_keys") - 800e6) < 5e6         # 640 MB of strided record reads + 160 MB of keys
    assert b.pmc_traffic(names, "no_such_kernel") is None
    assert b.pmc_profile("no_such_workload") == (None, None)             # counters belong to the profiled workload only
    names4, src4 = b.pmc_profile("cfg4")
    assert "commit" in src4 and b.pmc_traffic(names4, "k_hit_sub<gather>") > 1e10
