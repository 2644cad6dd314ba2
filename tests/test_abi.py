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
