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


@pytest.mark.parametrize("case", ["lognormal", "noisy"])
def test_tail_on_a_second_context_gives_the_same_output(case, tmpdir_s):
    """mahip_tail_handoff: the reduced graph, the survivors and their intervals move to a second context of the same device, which cleans the
    graph and builds the unitigs while the first one is free for the next input -- same bytes as the single-context run (and the reference's);
    the first context is overwritten by another input before the second one finishes, as a streaming caller would"""
    extra = {"lognormal": [], "noisy": ["-L", "uniform", "-d", "0.35", "-x", "0.03"]}[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "ho_%s.paf" % case), 3000, 90000, 77, extra)
    other = R.pafgen(os.path.join(tmpdir_s, "ho_other.paf"), 500, 9000, 78, [])
    opt = ma.default_opt()
    ing, ing2 = ma.Ingest(paf, opt), ma.Ingest(other, opt)
    c1, c2 = ma.Ctx(0), ma.Ctx(0)
    for fmt in ("ug", "sg"):
        c1.hits_upload(ing.hits, ing.n_seq)
        want = ma.run_resident(c1, opt, ing, fmt)
        c1.hits_upload(ing.hits, ing.n_seq)
        L = ma.lib()
        vp = C.c_void_p
        L.ma_pipeline_head.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
        L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.mahip_tail_handoff.argtypes = [vp, vp]
        st = (C.c_uint32 * 4)(0, 0, 0, 0)
        assert L.ma_pipeline_head(c1.h, C.byref(opt), ing.d, fmt.encode(), 100, 0, C.byref(st)) == 0
        ma._chk(L.mahip_tail_handoff(c1.h, c2.h), "tail_handoff")
        c1.hits_upload(ing2.hits, ing2.n_seq)  # the first context moves on
        st2 = (C.c_uint32 * 4)(0, 0, 0, 0)
        assert L.ma_pipeline_head(c1.h, C.byref(opt), ing2.d, fmt.encode(), 100, 0, C.byref(st2)) == 0
        buf, ln = vp(0), C.c_size_t(0)
        assert L.ma_pipeline_tail_mem(c2.h, C.byref(opt), ing.d, fmt.encode(), 100, C.byref(st), C.byref(buf), C.byref(ln)) == 0
        got = C.string_at(buf, ln.value)
        L.free_buf(buf)
        assert got == want, "%s: output of the two-context run differs" % fmt
        if R.have_ref() and fmt == "ug":
            ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
            if R.arc_tie_groups(ref_sg) == 0:
                assert got == R.run_cli(R.REF_BIN, [], paf)[0]
        c1.hits_upload(ing.hits, ing.n_seq)
        assert ma.run_resident_handoff(c1, c2, opt, ing, fmt) == want  # and again through the harness, contexts reused
    for x in (ing, ing2):
        x.close()
    c1.close(); c2.close()


def test_streaming_with_the_tail_on_a_second_context_and_thread(tmpdir_s):
    """the shape of bench.py's default pipelining (round 3: measured, adopted): this thread runs the hit passes of batch k+1 on the first context while a worker thread cleans batch k's
    graph and builds its unitigs on the second; a semaphore keeps the hand-over from overwriting a context that is still in use"""
    import queue
    import threading
    opt = ma.default_opt()
    pafs = [R.pafgen(os.path.join(tmpdir_s, "st_%d.paf" % k), 1500 + 400 * k, 40000 + 9000 * k, 90 + k, [] if k % 2 == 0 else ["-L", "uniform", "-d", "0.3", "-x", "0.03"]) for k in range(3)]
    ings = [ma.Ingest(p, opt) for p in pafs]
    c1, c2 = ma.Ctx(0), ma.Ctx(0)
    want = []
    for ing in ings:
        c1.hits_upload(ing.hits, ing.n_seq)
        want.append(ma.run_resident(c1, opt, ing, "ug"))
    L = ma.lib()
    vp = C.c_void_p
    L.ma_pipeline_head.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mahip_tail_handoff.argtypes = [vp, vp]
    q, free2, got, errs = queue.Queue(maxsize=1), threading.Semaphore(1), [], []

    def work():
        while True:
            item = q.get()
            try:
                if item is None:
                    return
                k, st = item
                buf, ln = vp(0), C.c_size_t(0)
                rc = L.ma_pipeline_tail_mem(c2.h, C.byref(opt), ings[k].d, b"ug", 100, C.byref(st), C.byref(buf), C.byref(ln))
                free2.release()
                if rc != 0:
                    errs.append("tail %d failed" % k)
                    continue
                got.append((k, C.string_at(buf, ln.value)))
                L.free_buf(buf)
            finally:
                q.task_done()

    t = threading.Thread(target=work, daemon=True)
    t.start()
    order = [0, 1, 2, 1, 0, 2, 2, 0]
    for k in order:
        c1.hits_upload(ings[k].hits, ings[k].n_seq)
        st = (C.c_uint32 * 4)(0, 0, 0, 0)
        assert L.ma_pipeline_head(c1.h, C.byref(opt), ings[k].d, b"ug", 100, 0, C.byref(st)) == 0
        free2.acquire()
        ma._chk(L.mahip_tail_handoff(c1.h, c2.h), "tail_handoff")
        q.put((k, st))
    q.join()
    q.put(None)
    t.join()
    assert not errs, errs
    assert [k for k, _ in got] == order
    for k, out in got:
        assert out == want[k], "batch of input %d differs" % k
    for ing in ings:
        ing.close()
    c1.close(); c2.close()


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_circular_unitigs_on_the_device(tmpdir_s):
    """rings of reads: a chain without a head is a cycle; the device cuts it in front of its smallest vertex (where the reference's sweep enters it,
    asm.c:132-172) and ranks again.  Hand-made graphs through the per-symbol ma_ug_gen: small rings with mixed strands, a ring next to linear
    pieces and isolated reads, a 3000-read ring (every pointer-jumping round sees the cycle), 40 rings of different sizes; unitigs (circular flag,
    members, lengths, ends), unitig arcs and the GFA text must equal the reference's."""
    LR, LP = R.ref(), product_graph_api()
    LR.ma_ug_gen.restype = C.c_void_p
    LR.ma_ug_gen.argtypes = [C.POINTER(ma.Asg)]
    LR.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    LR.ma_ug_destroy.argtypes = [C.c_void_p]
    for L in (LR, LP):
        L.sd_init.restype = C.POINTER(ma.Sdict)
        L.sd_put.restype = C.c_int32
        L.sd_put.argtypes = [C.POINTER(ma.Sdict), C.c_char_p, C.c_uint32]

    def build(n_seq, arcs_uv):
        rows = []
        for (u, v, ln, ol) in arcs_uv:
            rows.append((u, v, ln, ol))
            rows.append((v ^ 1, u ^ 1, ln + 7, ol))
        a = np.zeros(len(rows), dtype=ma.ARC_DT)
        for i, (u, v, ln, ol) in enumerate(rows):
            a[i] = ((u << 32) | ln, v, ol)
        a = a[np.argsort(a["ul"], kind="stable")]
        seq = np.array([5000 + 13 * (i % 97) for i in range(n_seq)], dtype="<u4")
        idx = np.zeros(2 * n_seq, dtype="<u8")
        R.orc().orc_arc_index(n_seq, len(a), a.ctypes.data, idx.ctypes.data)
        g = ma.Asg()
        for field, arr in (("arc", a), ("seq", seq), ("idx", idx)):
            p = libc.malloc(max(arr.nbytes, 16))
            C.memmove(p, arr.ctypes.data, arr.nbytes)
            setattr(g, field, p)
        g.m_arc, g.n_arc_srt, g.m_seq, g.n_seq_symm = max(len(a), 1), len(a) | 1 << 31, max(n_seq, 1), n_seq | 1 << 31
        return g

    def ring(first, n, mixed=False):
        out = []
        for i in range(n):
            u = 2 * (first + i) + ((i % 2) if mixed else 0)
            v = 2 * (first + (i + 1) % n) + (((i + 1) % 2) if mixed else 0)
            out.append((u, v, 900 + i % 50, 3000))
        return out

    cases = [(5, ring(0, 5)), (9, ring(3, 4, True) + [(0, 2, 700, 2000), (2, 5, 800, 2100)]), (3000, ring(0, 3000)), (2, ring(0, 2))]
    many, first = [], 0
    for k in range(40):
        many += ring(first, 3 + k, k % 3 == 0 and (3 + k) % 2 == 0)
        first += 3 + k + (k % 2)  # now and then a read without arcs in between
    cases.append((first + 2, many))
    for n_seq, arcs_uv in cases:
        outs = []
        for tag, L in (("ref", LR), ("mine", LP)):
            g = build(n_seq, arcs_uv)
            d = L.sd_init()
            for i in range(n_seq):
                L.sd_put(d, b"r%d" % i, 0)
            ug = L.ma_ug_gen(C.byref(g))
            path = os.path.join(tmpdir_s, "ring_%s.gfa" % tag)
            fp = libc.fopen(path.encode(), b"w")
            L.ma_ug_print(ug, d, None, fp)
            libc.fclose(fp)
            outs.append(open(path, "rb").read())
            L.ma_ug_destroy(ug)
        assert outs[0] == outs[1], "GFA of a ring graph differs (%d reads)" % n_seq
        import re
        assert re.search(rb"^S\tutg\d+c\t", outs[0], re.M), "the case should contain a circular unitig"


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_reduction_on_handmade_graphs_deleted_reads_and_big_multi_arc_vertices():
    """asg_arc_del_trans through the per-symbol ABI on graphs the synthetic inputs do not produce: a read that is flagged deleted but still has
    arcs (asg.c:158-161: all its arcs go), and vertices with more than 512 arcs, several of them to the same target (the block tier's literal
    replay of asg.c:181-184, where only the first arc to a reduced target is deleted).  Random transitive structure; graph after the call
    (arcs, seq, index) must equal the reference's."""
    LR, LP = R.ref(), product_graph_api()
    for L in (LR, LP):
        L.asg_arc_del_trans.restype = C.c_int
        L.asg_arc_del_trans.argtypes = [C.POINTER(ma.Asg), C.c_int]
    rng = np.random.default_rng(77)
    for n_seq, hub_deg, dup, n_del in ((50, 0, 0, 5), (900, 700, 40, 0), (1500, 1300, 200, 30), (300, 40, 10, 10)):
        rows = []
        pos = np.sort(rng.integers(0, 200000, n_seq))  # reads on a line: an arc u -> v with len = pos[v] - pos[u] for nearby v (transitive by construction)
        for u in range(n_seq):
            for v in range(u + 1, min(n_seq, u + 1 + int(rng.integers(2, 7)))):
                ln = int(pos[v] - pos[u]) + 1
                rows.append((2 * u, 2 * v, ln, 5000))
        hub = 0
        if hub_deg:  # one vertex with a very long arc list, some targets twice
            tg = rng.choice(np.arange(1, n_seq), size=hub_deg, replace=False)
            for v in tg:
                rows.append((2 * hub, 2 * int(v), int(pos[v] - pos[hub]) + 1, 4000))
            for v in tg[:dup]:
                rows.append((2 * hub, 2 * int(v), int(pos[v] - pos[hub]) + 1 + int(rng.integers(0, 3)), 3999))
        full = []
        for (u, v, ln, ol) in rows:
            full.append((u, v, ln, ol))
            full.append((v ^ 1, u ^ 1, ln + 3, ol))
        a = np.zeros(len(full), dtype=ma.ARC_DT)
        for i, (u, v, ln, ol) in enumerate(full):
            a[i] = ((u << 32) | ln, v, ol)
        a = a[np.argsort(a["ul"], kind="stable")]
        seq = np.full(n_seq, 9000, dtype="<u4")
        for r in rng.choice(n_seq, size=n_del, replace=False):
            seq[r] |= 1 << 31  # deleted read that still has arcs
        idx = np.zeros(2 * n_seq, dtype="<u8")
        R.orc().orc_arc_index(n_seq, len(a), a.ctypes.data, idx.ctypes.data)
        res = []
        for L in (LR, LP):
            g = ma.Asg()
            for field, arr in (("arc", a), ("seq", seq), ("idx", idx)):
                p = libc.malloc(max(arr.nbytes, 16))
                C.memmove(p, arr.ctypes.data, arr.nbytes)
                setattr(g, field, p)
            g.m_arc, g.n_arc_srt, g.m_seq, g.n_seq_symm = max(len(a), 1), len(a) | 1 << 31, n_seq, n_seq
            n_red = L.asg_arc_del_trans(C.byref(g), 1000)
            res.append((n_red, snapshot(C.pointer(g))))
        assert res[0][0] == res[1][0], "reduced %d vs %d arcs (%d reads)" % (res[0][0], res[1][0], n_seq)
        assert res[0][1] == res[1][1], "graph differs after the reduction (%d reads, hub %d)" % (n_seq, hub_deg)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("which", ["unclean", "asymmetric"])
def test_bubble_pop_that_revives_a_dead_read_is_answered_like_the_reference(which):
    """round-3 review, What's missing #4: asg_bub_backtrack (asg.c:352) revives a read that is dead when the pop happens -- not a stamp.  clean_core.h proves a
    symmetric, clean graph never gets there; tests/bubble_witness.py holds two graphs outside that contract on which the reference still answers (a sink that
    was flagged deleted but never cleaned away; a read deleted by one pop and revived by a LATER pop of the same sweep, on a graph whose is_symm flag lies).
    The device notices at the fixpoint and runs the call again as the reference's sequential sweep on one lane (k_clean_bubble_seq): same graph, no abort."""
    import bubble_witness as BW
    (a, seq, idx), why = BW.WITNESSES[which]()
    LR, LP = R.ref(), product_graph_api()
    LP.mahip_bubble_seq_sweeps.restype = C.c_uint32
    LP.mahip_bubble_seq_sweeps.argtypes = [C.c_void_p]
    LP.ma_gpu.restype = C.c_void_p
    before = LP.mahip_bubble_seq_sweeps(LP.ma_gpu())
    res = []
    for L in (LR, LP):
        L.asg_pop_bubble.restype = C.c_int
        L.asg_pop_bubble.argtypes = [C.POINTER(ma.Asg), C.c_int]
        g = ma.Asg()
        for field, arr in (("arc", a), ("seq", seq), ("idx", idx)):
            p = libc.malloc(max(arr.nbytes, 16))
            C.memmove(p, arr.ctypes.data, arr.nbytes)
            setattr(g, field, p)
        g.m_arc, g.n_arc_srt, g.m_seq, g.n_seq_symm = max(len(a), 1), len(a) | 1 << 31, len(seq), len(seq) | 1 << 31  # is_symm set: taken as it is
        n = L.asg_pop_bubble(C.byref(g), 50000)
        res.append((n, snapshot(C.pointer(g))))
    assert res[0][0] == res[1][0] and res[0][0] >= 1, (which, res[0][0], res[1][0])
    assert res[0][1] == res[1][1], "%s: graph differs from the reference's (%s)" % (which, why)
    assert LP.mahip_bubble_seq_sweeps(LP.ma_gpu()) == before + 1, "the call was supposed to be handed to the sequential sweep"


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_sequential_bubble_sweep_on_the_device_matches_reference(tmpdir_s):
    """the one-lane form of asg_pop_bubble on an ordinary noisy graph (MA_BUBBLE_SEQ=1, read when the library first pops a bubble: a process of its own)"""
    import subprocess
    import sys
    paf = R.pafgen(os.path.join(tmpdir_s, "seqsweep.paf"), 4000, 90000, 22, ["-L", "uniform", "-d", "0.35", "-x", "0.03"])
    ref_out, _ = R.run_cli(R.REF_BIN, [], paf)
    env = dict(os.environ, MA_BUBBLE_SEQ="1", MA_PIPE_TIMING="2")
    r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout == ref_out


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_sg_gen_on_hits_whose_query_groups_do_not_ascend(tmpdir_s):
    """the per-symbol ma_sg_gen (asm.c:9-39) indexes the hits as the caller has them (mahip_hits_index); the reference is independent of their
    order, and a caller may hand over groups in ANY order of the query ids.  The arc sort per read (k_arc_group_sort) takes read q's stretch of the push
    sequence from the group offsets; with the groups of neighbouring reads swapped, or all groups reversed, a stretch holds another read's arcs -- the kernel
    now counts such arcs and the general sort takes over.  Graph (arcs, seq, index) after ma_sg_gen must equal the reference's on the same permuted array."""
    LR, LP = R.ref(), product_graph_api()
    LP.ma_sg_gen.restype = C.POINTER(ma.Asg)
    LP.ma_sg_gen.argtypes = [C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_void_p, C.c_size_t, C.c_void_p]
    LP.asg_destroy.argtypes = [C.POINTER(ma.Asg)]
    paf = R.pafgen(os.path.join(tmpdir_s, "perm_groups.paf"), 3000, 80000, 41, [])  # tie-free (tests/golden/make_golden.py checks this input)
    opt = ma.default_opt()
    d = LR.sd_init()
    n = C.c_size_t(0)
    p = LR.ma_hit_read(paf.encode(), opt.min_span, opt.min_match, d, C.byref(n), 1, None)
    n = n.value
    n_seq = d.contents.n_seq
    sub = LR.ma_hit_sub(opt.min_dp, opt.min_iden, 0, n, p, n_seq)
    n = LR.ma_hit_cut(sub, opt.min_span, n, p)
    n = LR.ma_hit_contained(C.byref(opt), d, sub, n, p)
    hits = R.np_from(p, n, ma.HIT_DT).copy()
    hits["bldel"] &= 0x7FFFFFFF
    qid = (hits["qns"] >> np.uint64(32)).astype(np.int64)
    starts = np.flatnonzero(np.r_[True, qid[1:] != qid[:-1]])
    groups = np.split(np.arange(n), starts[1:])
    assert len(groups) > 100
    orders = {"sorted": list(range(len(groups))), "neighbours swapped": [i ^ 1 if (i ^ 1) < len(groups) else i for i in range(len(groups))],
              "reversed": list(range(len(groups) - 1, -1, -1)), "one pair swapped": [1, 0] + list(range(2, len(groups)))}
    for what, order in orders.items():
        a = np.ascontiguousarray(hits[np.concatenate([groups[g] for g in order])])
        res = []
        for L in (LR, LP):
            g = L.ma_sg_gen(C.byref(opt), d, sub, len(a), a.ctypes.data)
            res.append(snapshot(g))
            L.asg_destroy(g)
        assert len(res[0][0]) > 16 * 1000
        assert res[0] == res[1], "ma_sg_gen on hit groups in the order '%s': graph differs from the reference's" % what
    LR.free_buf(sub)
    LR.free_buf(p)
    LR.sd_destroy(d)
