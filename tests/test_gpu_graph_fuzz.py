"""Random string graphs through the reference's per-symbol graph interface: the device passes (csrc/clean.hip over clean_core.h, graph.hip, ug.hip)
against the unmodified reference library after EVERY call of a random script.  The graphs the pipeline produces have a particular shape (reads on a
line, overlaps sorted by length); here the shapes are arbitrary but legal -- symmetric, sorted, indexed -- with tangles, long tips, nested bubbles,
rings, equal overlap lengths and isolated reads, and the scripts use parameters the command line never does.  Runs on the GPU and on the CPU build
of the kernels (tests/emu)."""
import ctypes as C
import os

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
from test_host_vs_ref import product_graph_api, snapshot, libc

pytestmark = pytest.mark.gpu


def random_graph(rng, n_seq):
    """directed arcs (u, v, len, ol) between vertices read<<1|strand; the complement arc v^1 -> u^1 is added for each, so the graph is symmetric"""
    kind = rng.integers(0, 4)
    pos = np.sort(rng.integers(0, 40 * n_seq + 10, n_seq))
    rlen = rng.integers(300, 1500, n_seq)
    strand = rng.integers(0, 2, n_seq)
    rows = {}

    def add(a, b, ln, ol):
        u, v = 2 * a + int(strand[a]), 2 * b + int(strand[b])
        if a == b or (u, v) in rows or (v ^ 1, u ^ 1) in rows:
            return
        rows[(u, v)] = (int(ln), int(ol))
        rows[(v ^ 1, u ^ 1)] = (int(ln) + int(rng.integers(0, 3)), int(ol))

    width = int(rng.integers(1, 6))
    for a in range(n_seq):  # a backbone: every read overlaps a few of the next ones on the line
        for b in range(a + 1, min(n_seq, a + 1 + width)):
            if rng.random() < (0.9 if kind != 2 else 0.6):
                add(a, b, pos[b] - pos[a] + 1, max(1, rlen[a] - (pos[b] - pos[a])) if kind != 3 else 500)
    n_noise = int(rng.integers(0, n_seq // 2 + 2))
    for _ in range(n_noise):  # chords: tips, bubbles, tangles, back arcs (rings)
        a, b = int(rng.integers(0, n_seq)), int(rng.integers(0, n_seq))
        add(a, b, rng.integers(1, 3000), rng.integers(1, 1500))
    if not rows:
        add(0, min(1, n_seq - 1), 10, 10) if n_seq > 1 else None
    arcs = np.zeros(len(rows), dtype=ma.ARC_DT)
    for i, ((u, v), (ln, ol)) in enumerate(rows.items()):
        arcs[i] = ((u << 32) | ln, v, ol)
    arcs = arcs[np.argsort(arcs["ul"], kind="stable")]
    seq = rlen.astype("<u4")
    idx = np.zeros(2 * n_seq, dtype="<u8")
    R.orc().orc_arc_index(n_seq, len(arcs), arcs.ctypes.data, idx.ctypes.data)
    return arcs, seq, idx


def to_asg(arcs, seq, idx):
    g = ma.Asg()
    for field, arr in (("arc", arcs), ("seq", seq), ("idx", idx)):
        p = libc.malloc(max(arr.nbytes, 16))
        C.memmove(p, arr.ctypes.data, arr.nbytes)
        setattr(g, field, p)
    n_seq = len(seq)
    g.m_arc, g.n_arc_srt, g.m_seq, g.n_seq_symm = max(len(arcs), 1), len(arcs) | 1 << 31, max(n_seq, 1), n_seq | 1 << 31
    return g


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", range(8))
def test_random_graphs_random_scripts_match_reference_after_every_call(seed, tmpdir_s):
    LR, LP = R.ref(), product_graph_api()
    for L in (LR, LP):
        L.asg_arc_del_trans.restype = C.c_int
        L.asg_arc_del_trans.argtypes = [C.POINTER(ma.Asg), C.c_int]
        L.asg_symm.argtypes = [C.POINTER(ma.Asg)]
        L.sd_init.restype = C.POINTER(ma.Sdict)
        L.sd_put.restype = C.c_int32
        L.sd_put.argtypes = [C.POINTER(ma.Sdict), C.c_char_p, C.c_uint32]
        for name in ("asg_cut_tip", "asg_cut_internal", "asg_cut_biloop", "asg_pop_bubble"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(ma.Asg), C.c_int]
        L.asg_arc_del_short.restype = C.c_int
        L.asg_arc_del_short.argtypes = [C.POINTER(ma.Asg), C.c_float]
    LR.ma_ug_gen.restype = C.c_void_p
    LR.ma_ug_gen.argtypes = [C.POINTER(ma.Asg)]
    LR.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    LR.ma_ug_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(1000 + seed)
    for case in range(12):
        n_seq = int(rng.choice([2, 5, 12, 40, 150, 600]))
        arcs, seq, idx = random_graph(rng, n_seq)
        g_ref, g_mine = to_asg(arcs, seq, idx), to_asg(arcs, seq, idx)
        script = [("asg_arc_del_trans", int(rng.choice([0, 10, 1000])))] if rng.random() < 0.7 else []
        script.append(("asg_symm", None))  # the cleaners expect a symmetric graph without multi-arcs (main.c runs it right after the reduction)
        for _ in range(int(rng.integers(3, 9))):
            fn = str(rng.choice(["asg_cut_tip", "asg_pop_bubble", "asg_cut_internal", "asg_cut_biloop", "asg_arc_del_short"]))
            arg = {"asg_cut_tip": int(rng.integers(1, 7)), "asg_pop_bubble": int(rng.choice([100, 2000, 50000])), "asg_cut_internal": int(rng.integers(1, 4)),
                   "asg_cut_biloop": int(rng.integers(1, 7)), "asg_arc_del_short": float(np.float32(rng.choice([0.3, 0.5, 0.7, 0.9])))}[fn]
            script.append((fn, arg))
        for fn, arg in script:
            if fn == "asg_symm":
                LR.asg_symm(C.byref(g_ref)); LP.asg_symm(C.byref(g_mine))
                r0 = r1 = 0
            else:
                r0, r1 = getattr(LR, fn)(C.byref(g_ref), arg), getattr(LP, fn)(C.byref(g_mine), arg)
                if fn == "asg_arc_del_short" and r0:  # asg.c:95-98 symmetrises inside; keep both sides in step
                    pass
            assert r0 == r1, "seed %d case %d: %s(%r) returned %d vs %d" % (seed, case, fn, arg, r0, r1)
            assert snapshot(C.pointer(g_ref)) == snapshot(C.pointer(g_mine)), "seed %d case %d (%d reads): graph differs after %s(%r)" % (seed, case, n_seq, fn, arg)
        outs = []
        for tag, L, g in (("ref", LR, g_ref), ("mine", LP, g_mine)):
            d = L.sd_init()
            for i in range(n_seq):
                L.sd_put(d, b"r%d" % i, 0)
            ug = L.ma_ug_gen(C.byref(g))
            path = os.path.join(tmpdir_s, "gf_%d_%s.gfa" % (seed, tag))
            fp = libc.fopen(path.encode(), b"w")
            L.ma_ug_print(ug, d, None, fp)
            libc.fclose(fp)
            outs.append(open(path, "rb").read())
            L.ma_ug_destroy(ug)
        assert outs[0] == outs[1], "seed %d case %d: unitigs differ" % (seed, case)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_unitigs_of_asymmetric_graphs_through_the_per_symbol_abi(tmpdir_s):
    """ma_ug_gen on graphs without mirror arcs: the device sees that the links are not mirrored and runs the reference's sweep on one
    lane (k_ug_seq); unitigs may share reads.  Graphs the reference does not return on (found with the host build of the same walk) are left out."""
    import test_clean_core_cpu as TC
    LR, LP, LH = R.ref(), product_graph_api(), TC.host()
    vp, u32 = C.c_void_p, C.c_uint32
    LH.clh_ug2.argtypes = [u32, u32, vp, vp, vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), vp, vp, vp, vp, vp, vp, vp, C.c_size_t]
    for L in (LR, LP):
        L.sd_init.restype = C.POINTER(ma.Sdict)
        L.sd_put.restype = C.c_int32
        L.sd_put.argtypes = [C.POINTER(ma.Sdict), C.c_char_p, C.c_uint32]
        L.ma_ug_gen.restype = C.c_void_p
        L.ma_ug_gen.argtypes = [C.POINTER(ma.Asg)]
        L.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
        L.ma_ug_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(4242)
    n_done = n_overlap = 0
    for case in range(60):
        n_seq = int(rng.choice([3, 8, 30, 120]))
        arcs, seq, idx = TC.random_asym_graph(rng, n_seq, int(rng.integers(2, 2 * n_seq)), self_twin=bool(case & 1))
        V = 2 * n_seq
        cap = V * V + 16
        scratch = [np.zeros(V, dtype="<u4") for _ in range(5)]
        members, uarcs = np.zeros(cap + 1, dtype="<u8"), np.zeros(max(len(arcs), 1), dtype=ma.ARC_DT)
        n_utg, n_mem, n_ua = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        rc = LH.clh_ug2(n_seq, len(arcs), arcs.ctypes.data, idx.ctypes.data, seq.ctypes.data, C.byref(n_utg), C.byref(n_mem), C.byref(n_ua),
                        *[x.ctypes.data for x in scratch], members.ctypes.data, uarcs.ctypes.data, cap)
        if rc < 0:
            continue
        n_overlap += n_mem.value > V
        outs = []
        for tag, L in (("ref", LR), ("mine", LP)):
            g = to_asg(arcs, seq, idx)
            d = L.sd_init()
            for i in range(n_seq):
                L.sd_put(d, b"r%d" % i, 0)
            ug = L.ma_ug_gen(C.byref(g))
            path = os.path.join(tmpdir_s, "ga_%s.gfa" % tag)
            fp = libc.fopen(path.encode(), b"w")
            L.ma_ug_print(ug, d, None, fp)
            libc.fclose(fp)
            outs.append(open(path, "rb").read())
            L.ma_ug_destroy(ug)
        assert outs[0] == outs[1], "case %d (%d reads, %d arcs): unitigs differ" % (case, n_seq, len(arcs))
        n_done += 1
    assert n_done >= 40
