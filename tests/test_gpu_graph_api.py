"""The device graph passes behind the reference's per-symbol link interface (asg_cut_tip, asg_pop_bubble, asg_cut_internal,
asg_cut_biloop, asg_arc_del_short, ma_ug_gen: upload -> device -> download) against the unmodified reference library on the
GPU: the graph must equal the reference's after EVERY call of the cleaning script, then unitigs and GFA text byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
import stages as ST
from test_host_vs_ref import GRAPH_CASES, clone_graph, snapshot, product_graph_api, cleaning_script, libc

pytestmark = pytest.mark.gpu

CASES = GRAPH_CASES + [
    ("noisy_genome_order", 6000, 140000, 25, ["-g", "-L", "uniform", "-d", "0.4", "-x", "0.05"]),
    ("noisy_big", 60000, 1500000, 26, ["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,reads,lines,seed,extra", CASES, ids=[c[0] for c in CASES])
def test_device_cleaners_unitigs_gfa_match_reference(name, reads, lines, seed, extra, tmpdir_s):
    paf = R.pafgen(os.path.join(tmpdir_s, "ga_%s.paf" % name), reads, lines, seed, extra)
    opt = ma.default_opt()
    S = ST.ref_stages(paf, opt)
    LR, LP = R.ref(), product_graph_api()
    g_ref = S["g"]
    g_mine = clone_graph(g_ref)
    n_events = 0
    for fn, arg in cleaning_script(opt):
        if fn == "short":
            r0 = LR.asg_arc_del_short(g_ref, arg)
            r1 = LP.asg_arc_del_short(C.byref(g_mine), arg)
            if r0:  # reference main.c:169-172
                for f2, a2 in (("asg_cut_tip", opt.max_ext), ("asg_pop_bubble", opt.bub_dist)):
                    assert getattr(LR, f2)(g_ref, a2) == getattr(LP, f2)(C.byref(g_mine), a2)
        else:
            r0 = getattr(LR, fn)(g_ref, arg)
            r1 = getattr(LP, fn)(C.byref(g_mine), arg)
        assert r0 == r1, (fn, arg, r0, r1)
        n_events += r0 != 0
        assert snapshot(g_ref) == snapshot(C.pointer(g_mine)), "graph differs after %s(%r)" % (fn, arg)
    if name.startswith("noisy"):
        assert n_events >= 2, "noisy input should exercise the cleaners"
    d = LP.sd_init()
    dr = LR.sd_init()
    for i, nm in enumerate(S["names"]):
        assert LP.sd_put(d, nm.encode(), 0) == i
        LR.sd_put(dr, nm.encode(), 0)
    LR.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    sub = S["cont_sub"]
    ug_r, ug_p = LR.ma_ug_gen(g_ref), LP.ma_ug_gen(C.byref(g_mine))
    outs = []
    for tag, L, ug, dd in (("ref", LR, ug_r, dr), ("mine", LP, ug_p, d)):
        path = os.path.join(tmpdir_s, "ga_%s_%s.gfa" % (name, tag))
        fp = libc.fopen(path.encode(), b"w")
        L.ma_ug_print(ug, dd, sub.ctypes.data, fp)
        libc.fclose(fp)
        outs.append(open(path, "rb").read())
    assert outs[0] == outs[1], "GFA text differs (byte for byte, line order included)"
    assert outs[0].count(b"\nS\t") + outs[0].startswith(b"S\t") >= 1
    LR.ma_ug_destroy(ug_r); LP.ma_ug_destroy(ug_p)
    LR.asg_destroy(g_ref)
    LR.sd_destroy(dr); LP.sd_destroy(d)
