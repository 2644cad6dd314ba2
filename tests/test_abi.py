"""CPU: the C-ABI library loads and exports every symbol include/*.h declares, record layouts match the
reference's sizes, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import miniasm_amd as ma

INC = os.path.join(ma.ROOT, "include")
PREFIXES = ("mahip_", "ma_", "sd_", "paf_", "sys_", "asg_")


def declared_functions():
    names = set()
    for fn in os.listdir(INC):
        txt = open(os.path.join(INC, fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
        for stmt in txt.split(";"):
            if "typedef" in stmt or "{" in stmt:
                continue
            m = re.search(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", stmt)
            if m and m.group(1).startswith(PREFIXES):
                names.add(m.group(1))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    L = C.CDLL(ma.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 75, names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    assert C.c_int.in_dll(L, "ma_verbose").value == 3


def test_reference_externals_for_the_dropin_link():
    """symbols the reference's own main.o binds (nm -u oracle/_ref/main_ref.o), if that object was built"""
    obj = os.path.join(ma.ROOT, "oracle", "_ref", "main_ref.o")
    if not os.path.exists(obj):
        pytest.skip("oracle/_ref/main_ref.o not built")
    und = subprocess.run(["nm", "-u", obj], stdout=subprocess.PIPE, text=True).stdout.split()
    und = [s for s in und if s.startswith(PREFIXES)]
    L = C.CDLL(ma.LIB_PATH)
    assert len(und) >= 20
    assert not [s for s in und if not hasattr(L, s)]


def test_every_external_function_of_the_reference_objects_is_exported():
    """an object written against the reference's .o files may bind ANY of their non-static functions, declared in a header
    or not (paf_parse paf.c:34, sd_hash sdict.c:55, sys_liftrlimit sys.c:22 are not): all of them must resolve here"""
    ref = os.path.join(ma.ROOT, "oracle", "_ref", "libminiasm_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libminiasm_ref.so not built")
    out = subprocess.run(["nm", "-D", "--defined-only", ref], stdout=subprocess.PIPE, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TDB"]
    # klib macro instances (ks_*/kh_*/radix_sort_*/kdq_*) are static-inline-like helpers of the reference's containers, not its interface
    syms = [x for x in syms if x.startswith(PREFIXES) or x == "ma_verbose"]
    L = C.CDLL(ma.LIB_PATH)
    assert {"paf_parse", "sd_hash", "sys_liftrlimit"} <= set(syms)
    assert not [x for x in syms if not hasattr(L, x)]


def test_undeclared_reference_externals_behave_like_the_reference():
    L = C.CDLL(ma.LIB_PATH)

    class PafRec(C.Structure):  # paf.h:20-24
        _fields_ = [("qn", C.c_char_p), ("tn", C.c_char_p), ("ql", C.c_uint32), ("qs", C.c_uint32), ("qe", C.c_uint32),
                    ("tl", C.c_uint32), ("ts", C.c_uint32), ("te", C.c_uint32), ("ml_rev", C.c_uint32), ("bl", C.c_uint32)]
    libs = [L]
    ref = os.path.join(ma.ROOT, "oracle", "_ref", "libminiasm_ref.so")
    if os.path.exists(ref):
        libs.append(C.CDLL(ref))
    lines = [b"a\t9000\t10\t5000\t-\tb\t8000\t20\t5010\t800\t4990\t255", b"a\t9000\t10\t5000\t+\tb\t8000\t20\t5010\t800",
             b"a\t1\t2", b"q\t-5\t 7\t9x\t+\tt\t99999999999\t0\t0\t0\t0"]
    res = []
    for lib in libs:
        got = []
        for ln in lines:
            buf = C.create_string_buffer(ln, len(ln) + 1)
            r = PafRec(); r.bl = 77
            rc = lib.paf_parse(len(ln), buf, C.byref(r))
            got.append((rc, r.qn, r.tn, r.ql, r.qs, r.qe, r.tl, r.ts, r.te, r.ml_rev, r.bl) if rc >= 0 else (rc,))
        res.append(got)
    assert res[0][0] == (0, b"a", b"b", 9000, 10, 5000, 8000, 20, 5010, 800 | 1 << 31, 4990)
    assert res[0][1][-1] == 77 and res[0][2] == (-1,)  # 10 columns: bl untouched; 3 columns: refused
    assert all(r == res[0] for r in res)
    # sd_hash: the index exists afterwards, lookups work, a second call changes nothing
    L.sd_init.restype = C.c_void_p
    L.sd_put.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    L.sd_get.argtypes = [C.c_void_p, C.c_char_p]
    L.sd_hash.argtypes = [C.c_void_p]
    L.sd_destroy.argtypes = [C.c_void_p]
    d = L.sd_init()
    assert [L.sd_put(d, n, 5) for n in (b"x", b"y", b"x")] == [0, 1, 0]
    L.sd_hash(d); L.sd_hash(d)
    assert L.sd_get(d, b"y") == 1 and L.sd_get(d, b"z") == -1
    L.sd_destroy(d)
    L.sys_liftrlimit()


def test_record_layouts():
    assert C.sizeof(ma.MaOpt) == 56 and C.sizeof(ma.SdSeq) == 16 and C.sizeof(ma.Asg) == 40
    assert ma.HIT_DT.itemsize == 32 and ma.ARC_DT.itemsize == 16 and ma.SUB_DT.itemsize == 8
    o = ma.default_opt()
    assert (o.min_span, o.min_match, o.min_dp, o.max_hang, o.min_ovlp, o.gap_fuzz, o.n_rounds, o.bub_dist, o.max_ext) == (2000, 100, 3, 1000, 2000, 1000, 2, 50000, 4)
    assert abs(o.min_iden - .05) < 1e-7 and abs(o.int_frac - .8) < 1e-7
    L = ma.lib()
    L.ma_shard_stats_sizeof.restype = C.c_size_t
    assert L.ma_shard_stats_sizeof() == C.sizeof(ma.ShardStats), "miniasm_amd.ShardStats is out of step with host/ma_host.h: ma_shard_stats_t"
    assert len(ma.SHARD_PHASE_NAMES) == ma.SHARD_N_PHASES


def test_no_cpu_fallback(tmp_path):
    """without a GPU the library and the CLI must fail loudly, never compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ma.GpuError):
        ma.Ctx(0)
    paf = tmp_path / "x.paf"
    paf.write_text("a\t9000\t10\t5000\t+\tb\t9000\t20\t5010\t800\t4990\t255\n")
    r = subprocess.run([ma.CLI_PATH, str(paf)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"no CPU fallback" in r.stderr and r.stdout == b""


def test_product_does_not_link_the_oracle():
    out = subprocess.run(["ldd", ma.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "ma_oracle" not in out and "miniasm_ref" not in out
    for root, _, files in os.walk(os.path.join(ma.ROOT, "miniasm_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".hpp", ".py")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "ma_oracle" not in txt and "oracle/" not in txt.replace("CPU oracle", ""), f
