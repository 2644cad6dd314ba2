"""Host-side C of the product (PAF reader + dictionary, sequential graph cleaners, unitigs, GFA writer) against
the unmodified reference library, on the CPU.  These parts run on the host by design (they are text ingest or
sequential sweeps over the small reduced graph), so they can be pinned without a GPU."""
import ctypes as C
import gzip
import os
import shutil

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
import stages as ST

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]


def clone_graph(g):
    """deep copy of an asg_t into libc-malloc memory (both libraries free() their graphs)"""
    src = g.contents
    dst = ma.Asg()
    na, ns = src.n_arc, src.n_seq
    for field, nbytes in (("arc", max(na, 1) * 16), ("seq", max(ns, 1) * 4), ("idx", max(ns, 1) * 16)):
        p = libc.malloc(nbytes)
        C.memmove(p, getattr(src, field), nbytes if getattr(src, field) else 0)
        setattr(dst, field, p)
    dst.m_arc, dst.n_arc_srt = max(na, 1), src.n_arc_srt
    dst.m_seq, dst.n_seq_symm = max(ns, 1), src.n_seq_symm
    return dst


def snapshot(g):
    a, s, i = R.asg_arrays(g)
    return a.tobytes(), s.tobytes(), i.tobytes()


def product_graph_api():
    L = ma.lib()
    for name in ("asg_cut_tip", "asg_cut_internal", "asg_cut_biloop", "asg_pop_bubble"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(ma.Asg), C.c_int]
    L.asg_arc_del_short.restype = C.c_int
    L.asg_arc_del_short.argtypes = [C.POINTER(ma.Asg), C.c_float]
    L.ma_ug_gen.restype = C.c_void_p
    L.ma_ug_gen.argtypes = [C.POINTER(ma.Asg)]
    L.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    L.ma_sg_print.argtypes = [C.POINTER(ma.Asg), C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    L.ma_ug_destroy.argtypes = [C.c_void_p]
    L.sd_put.restype = C.c_int32
    L.sd_put.argtypes = [C.POINTER(ma.Sdict), C.c_char_p, C.c_uint32]
    return L


def cleaning_script(opt):
    """the call sequence of reference main.c:160-187 as (function, argument) pairs"""
    seq = [("asg_cut_tip", opt.max_ext), ("asg_pop_bubble", opt.bub_dist)]
    for i in range(opt.n_rounds + 1):
        r = np.float32(opt.min_ovlp_drop_ratio) + (np.float32(opt.max_ovlp_drop_ratio) - np.float32(opt.min_ovlp_drop_ratio)) / np.float32(opt.n_rounds) * np.float32(i)
        seq.append(("short", float(r)))
    seq += [("asg_cut_internal", 1), ("asg_cut_biloop", opt.max_ext), ("asg_cut_tip", opt.max_ext), ("asg_pop_bubble", opt.bub_dist),
            ("short", float(np.float32(opt.final_ovlp_drop_ratio)))]
    return seq


GRAPH_CASES = [
    ("clean", 1500, 40000, 21, []),
    ("noisy", 4000, 90000, 22, ["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    ("noisy2", 3000, 70000, 23, ["-L", "uniform", "-d", "0.5", "-x", "0.08"]),
    ("fixed", 2000, 50000, 24, ["-L", "fixed", "-d", "0.2", "-x", "0.02"]),
]


@needs_ref
@pytest.mark.parametrize("name,reads,lines,seed,extra", GRAPH_CASES, ids=[c[0] for c in GRAPH_CASES])
def test_text_writers_match_reference(name, reads, lines, seed, extra, tmpdir_s):
    """the host text writers (string-graph dump, unitig GFA incl. the multi-threaded / segmented formatting) on structures
    built by the reference: byte for byte the reference's text.  (The graph cleaners and the unitig construction run on the
    device: their algorithms are pinned on the CPU by tests/test_clean_core_cpu.py, the device code by tests/test_gpu_graph_api.py.)"""
    paf = R.pafgen(os.path.join(tmpdir_s, "h_%s.paf" % name), reads, lines, seed, extra)
    opt = ma.default_opt()
    S = ST.ref_stages(paf, opt)
    LR, LP = R.ref(), product_graph_api()
    g_ref = S["g"]
    for fn, arg in cleaning_script(opt):
        if fn == "short":
            if LR.asg_arc_del_short(g_ref, arg):
                LR.asg_cut_tip(g_ref, opt.max_ext); LR.asg_pop_bubble(g_ref, opt.bub_dist)
        else:
            getattr(LR, fn)(g_ref, arg)
    d = LP.sd_init()
    dr = LR.sd_init()
    for i, nm in enumerate(S["names"]):
        assert LP.sd_put(d, nm.encode(), 0) == i
        LR.sd_put(dr, nm.encode(), 0)
    LR.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    LR.ma_sg_print.argtypes = [C.POINTER(ma.Asg), C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    sub = S["cont_sub"]
    sg_txt = []
    for tag, L, dd in (("ref", LR, dr), ("mine", LP, d)):
        for with_sub in (True, False):
            path = os.path.join(tmpdir_s, "h_%s_%s_%d.sg" % (name, tag, with_sub))
            fp = libc.fopen(path.encode(), b"w")
            L.ma_sg_print(g_ref, dd, sub.ctypes.data if with_sub else None, fp)
            libc.fclose(fp)
            sg_txt.append(open(path, "rb").read())
    assert sg_txt[0] == sg_txt[2] and sg_txt[1] == sg_txt[3], "string-graph text differs"
    assert len(sg_txt[0]) > 100
    ug = LR.ma_ug_gen(g_ref)
    outs = []
    for tag, L, dd in (("ref", LR, dr), ("mine", LP, d)):
        path = os.path.join(tmpdir_s, "h_%s_%s.gfa" % (name, tag))
        fp = libc.fopen(path.encode(), b"w")
        L.ma_ug_print(ug, dd, sub.ctypes.data, fp)
        libc.fclose(fp)
        outs.append(open(path, "rb").read())
    assert outs[0] == outs[1], "GFA text differs (byte for byte, line order included)"
    assert outs[0].count(b"\nS\t") + outs[0].startswith(b"S\t") >= 1
    for grain, threads, seg in (("40", "7", "5"), ("3", "16", "1"), ("100000", "4", "3"), ("10", "3", "100000")):  # the writer formats big outputs on several threads, long unitigs in pieces: same bytes
        os.environ["MA_FMT_GRAIN"], os.environ["MA_THREADS"], os.environ["MA_FMT_SEG"] = grain, threads, seg
        try:
            path = os.path.join(tmpdir_s, "h_%s_mt.gfa" % name)
            fp = libc.fopen(path.encode(), b"w")
            LP.ma_ug_print(ug, d, sub.ctypes.data, fp)
            libc.fclose(fp)
            assert open(path, "rb").read() == outs[0], "multi-threaded GFA text differs (grain %s)" % grain
        finally:
            del os.environ["MA_FMT_GRAIN"], os.environ["MA_THREADS"], os.environ["MA_FMT_SEG"]
    LR.ma_ug_destroy(ug)
    LR.asg_destroy(g_ref)
    LR.sd_destroy(dr); LP.sd_destroy(d)


@needs_ref
def test_paf_reader_edge_cases(tmpdir_s):
    """gz input, CRLF, short lines, 10-column lines (stale bl), junk numbers, no trailing newline"""
    lines = [
        b"a\t9000\t10\t5000\t+\tb\t9000\t20\t5010\t800\t4990\t255",
        b"a\t9000\t100\t6000\t-\tc\t9500\t0\t5900\t900\t5900\t255\r",
        b"short\tline",
        b"",
        b"b\t9000\t0\t4000\t+\tc\t9500\t5000\t9000\t700",          # 10 columns: bl keeps the previous value
        b"c\t9500\t+12\t 4000\t-\td\t8000x\t0\t3988\t600\t3988\t255",  # strtol-isms: sign, blank, trailing junk
        b"d\t8000\t0\t3000\t+\td\t8000\t10\t3010\t500\t3000\t255",   # self hit
        b"e\t7000\t0\t2500\t+\ta\t9000\t6500\t9000\t400\t2500\t255\textra\tcols",
        b"f\t7000\t0\t1500\t+\ta\t9000\t0\t1500\t400\t1500\t255",    # below min_span
        b"g\t7000\t0\t2500\t+\ta\t9000\t0\t2500\t40\t2500\t255",     # below min_match
        b"a\t9000\t3000\t8000\t+\tg\t7000\t0\t5000\t900\t5000\t255",  # no trailing newline below
    ]
    txt = b"\n".join(lines)
    paths = [os.path.join(tmpdir_s, "edge.paf"), os.path.join(tmpdir_s, "edge.paf.gz"), os.path.join(tmpdir_s, "edge_nl.paf")]
    open(paths[0], "wb").write(txt)
    with gzip.open(paths[1], "wb") as f:
        f.write(txt)
    open(paths[2], "wb").write(txt + b"\n")
    opt = ma.default_opt()
    LR = R.ref()
    for p in paths:
        ing = ma.Ingest(p, opt)
        d = LR.sd_init()
        n = C.c_size_t(0)
        q = LR.ma_hit_read(p.encode(), opt.min_span, opt.min_match, d, C.byref(n), 1, None)
        ref_hits = R.np_from(q, n.value, ma.HIT_DT)
        ref_hits["bldel"] &= 0x7FFFFFFF
        assert n.value == ing.n and n.value > 0
        assert R.canon(ref_hits).tobytes() == R.canon(ing.hits).tobytes(), p
        assert [d.contents.seq[i].name.decode() for i in range(d.contents.n_seq)] == ing.names()
        assert [d.contents.seq[i].len for i in range(d.contents.n_seq)] == list(ing.lens())
        LR.free_buf(q); LR.sd_destroy(d); ing.close()


@needs_ref
def test_ingest_large_lines_and_chunk_boundaries(tmpdir_s):
    """a file larger than the 1 MiB read chunk with long read names so that lines straddle chunk boundaries"""
    paf = R.pafgen(os.path.join(tmpdir_s, "chunk.paf"), 3000, 60000, 31, [])
    big = os.path.join(tmpdir_s, "chunk_long.paf")
    with open(paf, "rb") as f, open(big, "wb") as g:
        for ln in f:
            g.write(ln.replace(b"r", b"read_with_a_rather_long_name_" * 3))
    assert os.path.getsize(big) > 3 * (1 << 20)
    opt = ma.default_opt()
    ing = ma.Ingest(big, opt)
    LR = R.ref()
    d = LR.sd_init()
    n = C.c_size_t(0)
    q = LR.ma_hit_read(big.encode(), opt.min_span, opt.min_match, d, C.byref(n), 1, None)
    ref_hits = R.np_from(q, n.value, ma.HIT_DT)
    ref_hits["bldel"] &= 0x7FFFFFFF
    assert n.value == ing.n
    assert R.canon(ref_hits).tobytes() == R.canon(ing.hits).tobytes()
    assert [d.contents.seq[i].name.decode() for i in range(d.contents.n_seq)] == ing.names()
    LR.free_buf(q); LR.sd_destroy(d); ing.close()


def test_refsort_emulation_matches_reference_on_ties():
    """host asg_arc_sort (used for the unitig graph) reproduces the reference's unstable radix order, ties included"""
    if not R.have_ref():
        pytest.skip("oracle/_ref not built")
    LR, LP = R.ref(), ma.lib()
    LR.asg_arc_sort.argtypes = [C.POINTER(ma.Asg)]
    LP.asg_arc_sort.argtypes = [C.POINTER(ma.Asg)]
    rng = np.random.default_rng(5)
    for n, nu, nl in ((50, 4, 5), (300, 8, 6), (5000, 40, 9), (70000, 300, 30), (20000, 3, 2000)):
        arcs = np.zeros(n, dtype=ma.ARC_DT)
        arcs["ul"] = (rng.integers(0, nu, n).astype(np.uint64) << 32) | rng.integers(0, nl, n).astype(np.uint64)
        arcs["v"] = np.arange(n)  # distinguishes tied records
        res = []
        for L in (LR, LP):
            a = arcs.copy()
            g = ma.Asg()
            g.arc, g.n_arc_srt, g.m_arc = a.ctypes.data, n, n
            L.asg_arc_sort(C.byref(g))
            res.append(a.tobytes())
        assert res[0] == res[1], "tie order differs for n=%d" % n


def packed_order(LP, keys):
    """the order through ma_refsort_packed -- elements packed by the caller the way the device does it (csrc/radix.hip: k_pack_keys); None: keys too wide"""
    n = len(keys)
    LP.ma_refsort_packed.restype = C.c_int
    LP.ma_refsort_packed.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    hi, lo = keys >> np.uint64(32), keys & np.uint64(0xffffffff)
    bh, bl, bi = max(int(hi.max()).bit_length(), 1), max(int(lo.max()).bit_length(), 1), max((n - 1).bit_length(), 1)
    idx = np.arange(n, dtype=np.uint64)
    if bh + bl + bi <= 64:
        pk = np.ascontiguousarray(((hi << np.uint64(bl) | lo) << np.uint64(bi)) | idx)
        assert LP.ma_refsort_packed(pk.ctypes.data, n, bl, bi, -1, None) == 0
    else:
        st = (32 + bh - 1) & ~7
        if (st - 32) + bl + bi > 64 or n <= 64:
            return None
        pk = np.ascontiguousarray((((hi & np.uint64((1 << (st - 32)) - 1)) << np.uint64(bl) | lo) << np.uint64(bi)) | idx)
        dig = np.zeros(n + 16, dtype=np.uint8)
        dig[:n] = ((keys >> np.uint64(st)) & np.uint64(0xff)).astype(np.uint8)
        assert LP.ma_refsort_packed(pk.ctypes.data, n, bl, bi, st, dig.ctypes.data) == 0
    return (pk & np.uint64((1 << bi) - 1)).astype(np.uint32)


def test_refsort_packed_rejects_what_it_cannot_sort():
    """ma_refsort_packed's contract (host/refsort.c): bit counts in range, the top digit apart only with its digit array, at a multiple of 8 at or above
    bit 32, with room for the rest in the word, and never for <= 64 records (the reference sorts those by insertion: ksort.h:182)"""
    LP = ma.lib()
    LP.ma_refsort_packed.restype = C.c_int
    LP.ma_refsort_packed.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    pk = np.arange(1000, dtype=np.uint64)[::-1].copy()
    dig = np.zeros(1016, dtype=np.uint8)
    assert LP.ma_refsort_packed(pk.ctypes.data, 1, 10, 10, -1, None) == 0 and LP.ma_refsort_packed(pk.ctypes.data, 0, 10, 10, -1, None) == 0
    for bl, bi, st, d in ((0, 10, -1, None), (33, 10, -1, None), (10, 0, -1, None), (10, 33, -1, None),  # bit counts
                          (10, 10, 40, None), (10, 10, 24, dig), (10, 10, 36, dig), (32, 32, 40, dig)):  # no digits / below bit 32 / not a multiple of 8 / no room
        assert LP.ma_refsort_packed(pk.ctypes.data, 1000, bl, bi, st, d.ctypes.data if d is not None else None) == -1, (bl, bi, st)
    assert LP.ma_refsort_packed(pk.ctypes.data, 64, 10, 10, 40, dig.ctypes.data) == -1
    assert (pk == np.arange(1000, dtype=np.uint64)[::-1]).all(), "a rejected call must not touch the array"
    keys = np.arange(1000, dtype=np.uint64)[::-1].copy()  # (and one that is accepted: key above position, 10 + 10 bits)
    pk = (keys << np.uint64(10)) | np.arange(1000, dtype=np.uint64)
    assert LP.ma_refsort_packed(pk.ctypes.data, 1000, 10, 10, -1, None) == 0
    assert ((pk >> np.uint64(10)) == np.arange(1000, dtype=np.uint64)).all()


@needs_ref
@pytest.mark.parametrize("threads", ["1", "8"])
def test_refsort_perm_matches_reference_sort(threads, monkeypatch):
    """exact-tie mode: the permutation computed from the keys alone (host/refsort.c, sub-buckets on threads) is the
    one the reference's radix_sort_hit (hit.c:13,21) / asg_arc_sort (asg.c:9,24) apply to the records"""
    monkeypatch.setenv("MA_THREADS", threads)
    LR, LP = R.ref(), ma.lib()
    LP.ma_refsort_perm.restype = C.c_int
    LP.ma_refsort_perm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    LR.radix_sort_hit.argtypes = [C.c_void_p, C.c_void_p]
    LR.radix_sort_hit.restype = None
    rng = np.random.default_rng(17)
    # (n, reads, distinct starts): from the insertion-sort-only case to multi-level buckets with heavy ties
    for n, nq, ns in ((1, 1, 1), (64, 3, 4), (65, 3, 4), (1000, 1, 1), (3000, 7, 40), (200000, 300, 50), (700000, 70000, 3), (900000, 300000, 100000)):
        hits = np.zeros(n, dtype=ma.HIT_DT)
        hits["qns"] = (rng.integers(0, nq, n).astype(np.uint64) << 32) | (rng.integers(0, ns, n).astype(np.uint64) * 977 % 70001)
        hits["tn"] = np.arange(n)  # distinguishes tied records
        perm = np.zeros(n, dtype=np.uint32)
        keys = np.ascontiguousarray(hits["qns"])
        assert LP.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        ref = hits.copy()
        LR.radix_sort_hit(ref.ctypes.data, ref.ctypes.data + n * 32)
        assert (np.diff(ref["qns"].astype(np.int64)) >= 0).all()
        assert hits[perm].tobytes() == ref.tobytes(), "hit order differs for n=%d" % n
        pp = packed_order(LP, keys)
        assert pp is not None and (pp == perm).all(), "caller-packed elements: different order for n=%d" % n
    # the shape of a PAF file: keys that are almost sorted already (a query's overlaps are listed together, the mirrored records point at reads
    # nearby): the top-level walk (host/refsort_body.h: permute_top) then runs over long stretches of elements that are already home -- identity at
    # the bucket it works on, shifted by one everywhere else, recorded as segments when long
    # (the 2.6 M-element cases leave ranges of more than a million elements below the top level: those walk on a byte array of their digits too)
    for n, per, spread in ((300000, 3, 40), (1500000, 20, 3), (1500000, 20, 70000), (2000000, 7, 1), (2600000, 20, 500), (2600000, 13, 100000)):
        nr = n // per
        q = np.sort(rng.integers(0, nr, n // 2)).astype(np.int64)
        t = np.clip(q + rng.integers(-spread, spread + 1, n // 2), 0, nr - 1)
        qid = np.empty(n, dtype=np.uint64)
        qid[0::2], qid[1::2] = q.astype(np.uint64), t.astype(np.uint64)
        hits = np.zeros(n, dtype=ma.HIT_DT)
        hits["qns"] = (qid << np.uint64(32)) | (rng.integers(0, 50, n).astype(np.uint64) * 16)
        hits["tn"] = np.arange(n)
        perm = np.zeros(n, dtype=np.uint32)
        keys = np.ascontiguousarray(hits["qns"])
        assert LP.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        ref = hits.copy()
        LR.radix_sort_hit(ref.ctypes.data, ref.ctypes.data + n * 32)
        assert hits[perm].tobytes() == ref.tobytes(), "hit order differs for the nearly sorted input n=%d per=%d spread=%d" % (n, per, spread)
        pp = packed_order(LP, keys)
        assert pp is not None and (pp == perm).all(), "caller-packed elements: different order for the nearly sorted input n=%d" % n
    # keys + index too wide for one 64-bit element (BASELINE configs[4]: 23 + 14 + 30 bits): packed without the top level's digit when the rest fits
    # (30 id bits: top byte = id >> 24, 24 + 20 + 19 bits travel), the 16-byte elements when it does not (32 + 32 + 19)
    for n, hi_bits, lo_bits, n_hi in ((300000, 30, 20, 5000), (300000, 30, 20, 40), (400000, 32, 32, 3000)):
        pool = rng.integers(0, 1 << hi_bits, n_hi, dtype=np.uint64)
        hits = np.zeros(n, dtype=ma.HIT_DT)
        hits["qns"] = (pool[rng.integers(0, n_hi, n)] << np.uint64(32)) | (rng.integers(0, 64, n).astype(np.uint64) * np.uint64(((1 << lo_bits) - 1) // 64))
        hits["tn"] = np.arange(n)
        perm = np.zeros(n, dtype=np.uint32)
        keys = np.ascontiguousarray(hits["qns"])
        assert LP.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        ref = hits.copy()
        LR.radix_sort_hit(ref.ctypes.data, ref.ctypes.data + n * 32)
        assert hits[perm].tobytes() == ref.tobytes(), "hit order differs for wide keys (%d id bits, %d start bits)" % (hi_bits, lo_bits)
        pp = packed_order(LP, keys)
        assert (pp is None) == (hi_bits == 32) and (pp is None or (pp == perm).all()), "caller-packed wide keys (%d id bits): different order" % hi_bits
    LR.asg_arc_sort.argtypes = [C.POINTER(ma.Asg)]
    for n, nu, nl in ((300, 8, 6), (400000, 5000, 12), (600000, 400000, 3)):
        arcs = np.zeros(n, dtype=ma.ARC_DT)
        arcs["ul"] = (rng.integers(0, nu, n).astype(np.uint64) << 32) | rng.integers(0, nl, n).astype(np.uint64)
        arcs["v"] = np.arange(n)
        perm = np.zeros(n, dtype=np.uint32)
        keys = np.ascontiguousarray(arcs["ul"])
        assert LP.ma_refsort_perm(keys.ctypes.data, n, perm.ctypes.data) == 0
        ref = arcs.copy()
        g = ma.Asg()
        g.arc, g.n_arc_srt, g.m_arc = ref.ctypes.data, n, n
        LR.asg_arc_sort(C.byref(g))
        assert arcs[perm].tobytes() == ref.tobytes(), "arc order differs for n=%d" % n
        pp = packed_order(LP, keys)
        assert pp is not None and (pp == perm).all(), "caller-packed arc keys: different order for n=%d" % n


@needs_ref
def test_parallel_ingest_matches_reference(tmpdir_s, monkeypatch):
    """chunk-parallel ingest (ingest_mt.c): same ids, same records as the reference's sequential reader, including
    the `bl` a 10-column line inherits across a chunk cut, CRLF line ends and junk lines"""
    import random
    paf = R.pafgen(os.path.join(tmpdir_s, "mt.paf"), 6000, 260000, 33, ["-L", "uniform", "-x", "0.02"])
    rnd = random.Random(5)
    big = os.path.join(tmpdir_s, "mt_mod.paf")
    with open(paf, "rb") as f, open(big, "wb") as g:
        for ln in f:
            x = rnd.random()
            if x < 0.02:
                ln = b"\t".join(ln.rstrip(b"\n").split(b"\t")[:10]) + b"\n"      # 10 columns: inherits bl
            elif x < 0.03:
                ln = ln.rstrip(b"\n") + b"\r\n"
            elif x < 0.032:
                ln = b"junk\tline\n"
            g.write(ln)
    assert os.path.getsize(big) > (8 << 20)
    opt = ma.default_opt()
    LR = R.ref()
    d = LR.sd_init()
    n = C.c_size_t(0)
    q = LR.ma_hit_read(big.encode(), opt.min_span, opt.min_match, d, C.byref(n), 1, None)
    ref_hits = R.np_from(q, n.value, ma.HIT_DT)
    ref_hits["bldel"] &= 0x7FFFFFFF
    ref_names = [d.contents.seq[i].name.decode() for i in range(d.contents.n_seq)]
    ref_lens = [d.contents.seq[i].len for i in range(d.contents.n_seq)]
    ref_c = R.canon(ref_hits).tobytes()
    first = None
    for th in ("1", "3", "7", "16"):
        monkeypatch.setenv("MA_THREADS", th)
        ing = ma.Ingest(big, opt)
        assert ing.n == n.value, th
        assert ing.names() == ref_names and list(ing.lens()) == ref_lens, "dictionary differs with %s threads" % th
        assert R.canon(ing.hits).tobytes() == ref_c, "records differ with %s threads" % th
        raw = ing.hits.tobytes()
        if first is None:
            first = raw
        assert raw == first, "unsorted record order differs from the sequential path with %s threads" % th
        ing.close()
    LR.free_buf(q); LR.sd_destroy(d)


@needs_ref
@pytest.mark.gpu
def test_unitig_sequences_match_reference(tmpdir_s):
    """ma_ug_seq (-f): FASTA and FASTQ (gz, multi-line, CRLF, lower case, IUPAC, reads not in the graph).  The record reader is host
    code, the placement of the bases a device byte gather (csrc/useq.hip): needs the GPU."""
    import random
    paf = R.pafgen(os.path.join(tmpdir_s, "seq.paf"), 1500, 40000, 27, ["-L", "uniform", "-d", "0.3", "-x", "0.03"])
    opt = ma.default_opt()
    S = ST.ref_stages(paf, opt)
    LR, LP = R.ref(), product_graph_api()
    for L in (LR, LP):
        L.ma_ug_seq.restype = C.c_int
        L.ma_ug_seq.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_char_p]
        L.ma_ug_print.argtypes = [C.c_void_p, C.POINTER(ma.Sdict), C.c_void_p, C.c_void_p]
    # read lengths: every surviving read needs at least sub.e bases
    rnd = random.Random(3)
    need = {nm: int(s["e"]) for nm, s in zip(S["names"], S["cont_sub"])}
    alphabet = "ACGTacgtNnRYKMSWBDHV"
    fa, fq = os.path.join(tmpdir_s, "reads.fa"), os.path.join(tmpdir_s, "reads.fq.gz")
    recs = [("ghost1", "ACGT" * 10)]
    for nm, n in need.items():
        recs.append((nm, "".join(rnd.choice(alphabet) for _ in range(n + rnd.randrange(0, 50)))))
    recs.append(("ghost2", "TTTT"))
    with open(fa, "w") as f:
        for k, (nm, sq) in enumerate(recs):
            f.write(">%s some comment\n" % nm)
            w = 60 if k % 3 else 10 ** 9
            eol = "\r\n" if k % 5 == 0 else "\n"
            for i in range(0, len(sq), w):
                f.write(sq[i:i + w] + eol)
    with gzip.open(fq, "wt") as f:
        for nm, sq in recs:
            f.write("@%s\n%s\n+\n%s\n" % (nm, sq, "I" * len(sq)))
    sub = S["cont_sub"]
    for reads in (fa, fq):
        texts = []
        for L in (LR, LP):
            g = clone_graph(S["g"])
            d = L.sd_init()
            for i, nm in enumerate(S["names"]):
                L.sd_put(d, nm.encode(), 0)
            ug = LR.ma_ug_gen(C.byref(g))  # the unitigs themselves come from the device in the product (tests/test_gpu_graph_api.py)
            assert L.ma_ug_seq(ug, d, sub.ctypes.data, reads.encode()) == 0
            path = os.path.join(tmpdir_s, "seq_%d.gfa" % len(texts))
            fp = libc.fopen(path.encode(), b"w")
            L.ma_ug_print(ug, d, sub.ctypes.data, fp)
            libc.fclose(fp)
            texts.append(open(path, "rb").read())
            LR.ma_ug_destroy(ug); L.sd_destroy(d)
            for p in (g.arc, g.seq, g.idx):
                libc.free(C.c_void_p(p))
        assert texts[0] == texts[1], "unitig sequences differ (%s)" % reads
        first = texts[0].split(b"\n")[0].split(b"\t")
        assert first[0] == b"S" and first[2] != b"*" and len(first[2]) > 1000
    LR.asg_destroy(S["g"])


@needs_ref
def test_host_reader_fuzz_matches_reference(tmpdir_s):
    """random ASCII soup (separators, signs, blanks, CR, NUL over-represented) through the reference's ma_hit_read and through
    the host reader: same dictionary, same records.  (The device parser is fuzzed against the host reader in test_gpu_ingest.)"""
    import random
    rnd = random.Random(77)
    alphabet = "0123456789" * 6 + "\t" * 14 + "\n" * 3 + "+- \r\x00\x0b\x0cabcxyzACGT:_.|" + "\t\t"
    opt = ma.default_opt(); opt.min_span = 0; opt.min_match = 0
    LR = R.ref()
    for k in range(10):
        n = rnd.choice((0, 7, 300, 5000, 60000, 300000))
        txt = "".join(rnd.choice(alphabet) for _ in range(n))
        good = ["r%d\t9000\t%d\t%d\t%s\tr%d\t8000\t%d\t%d\t%d\t%d\t255" % (rnd.randint(0, 40), a, a + rnd.randint(0, 5000), rnd.choice("+-"), rnd.randint(0, 40), b,
                                                                     b + rnd.randint(0, 5000), rnd.randint(0, 900), rnd.randint(0, 4000))
                for a, b in ((rnd.randint(0, 3000), rnd.randint(0, 3000)) for _ in range(200))]
        parts = txt.split("\n")
        for g in good[1:]:
            parts.insert(rnd.randint(0, len(parts)), g)
        txt = good[0] + "\n" + "\n".join(parts)  # a full first line: the reference's `bl` is uninitialised before the first 11-column line
        p = os.path.join(tmpdir_s, "fz%d.paf" % k)
        open(p, "wb").write(txt.encode("ascii"))
        d = LR.sd_init()
        cnt = C.c_size_t(0)
        q = LR.ma_hit_read(p.encode(), opt.min_span, opt.min_match, d, C.byref(cnt), 1, None)
        ref_hits = R.np_from(q, cnt.value, ma.HIT_DT)
        ref_hits["bldel"] &= 0x7FFFFFFF
        ing = ma.Ingest(p, opt)
        assert ing.n == cnt.value, k
        assert [x.encode() for x in ing.names()] == [d.contents.seq[i].name for i in range(d.contents.n_seq)], k
        assert list(ing.lens()) == [d.contents.seq[i].len for i in range(d.contents.n_seq)], k
        assert R.canon(ing.hits).tobytes() == R.canon(ref_hits).tobytes(), k
        LR.free_buf(q); LR.sd_destroy(d); ing.close()


def test_dictionary_bulk_fill_arena_semantics():
    """ma_sd_fill (what the device-side ingest uses): names live in one block owned by the dictionary; the index is built on
    first use; sd_put adds ordinary names next to the arena ones; sd_squeeze and sd_destroy free each kind correctly"""
    L = ma.lib()
    L.ma_sd_fill.argtypes = [C.POINTER(ma.Sdict), C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    L.ma_sd_fill.restype = None
    L.sd_get.restype = C.c_int32
    L.sd_get.argtypes = [C.POINTER(ma.Sdict), C.c_char_p]
    L.sd_put.restype = C.c_int32
    L.sd_put.argtypes = [C.POINTER(ma.Sdict), C.c_char_p, C.c_uint32]
    L.sd_squeeze.restype = C.c_void_p
    L.sd_squeeze.argtypes = [C.POINTER(ma.Sdict)]
    names = [("read%d" % i).encode() for i in range(5000)]
    blob = b"\0".join(names) + b"\0"
    arena = libc.malloc(len(blob))
    C.memmove(arena, blob, len(blob))
    lens = np.arange(5000, dtype=np.uint32) + 100
    for refill in (False, True):
        d = L.sd_init()
        assert L.sd_put(d, b"old_name", 7) == 0  # a dictionary that already holds something is replaced by the fill
        if refill:  # fill twice: the first arena must be released by the second fill
            a0 = libc.malloc(len(blob)); C.memmove(a0, blob, len(blob))
            L.ma_sd_fill(d, a0, len(blob), 5000, lens.ctypes.data)
        a1 = libc.malloc(len(blob)); C.memmove(a1, blob, len(blob))
        L.ma_sd_fill(d, a1, len(blob), 5000, lens.ctypes.data)
        assert d.contents.n_seq == 5000 and d.contents.seq[4999].name == b"read4999" and d.contents.seq[17].len == 117
        assert L.sd_get(d, b"read1234") == 1234 and L.sd_get(d, b"old_name") == -1 and L.sd_get(d, b"nope") == -1
        assert L.sd_put(d, b"read42", 1) == 42            # known name: first length wins
        assert d.contents.seq[42].len == 142
        assert L.sd_put(d, b"late_comer", 9) == 5000       # strdup'ed next to the arena names
        for i in range(0, 5001, 2):
            d.contents.seq[i].auxdel |= 0x80000000         # drop every other read (and the late comer)
        m = L.sd_squeeze(d)
        mp = np.frombuffer(C.string_at(m, 5001 * 4), dtype=np.int32)
        L.free_buf(m)
        assert d.contents.n_seq == 2500 and mp[1] == 0 and mp[0] == -1 and mp[4999] == 2499 and mp[5000] == -1
        assert L.sd_get(d, b"read4999") == 2499 and L.sd_get(d, b"read4998") == -1 and L.sd_get(d, b"late_comer") == -1
        L.sd_destroy(d)
    libc.free(C.c_void_p(arena))


def test_cpu_budget_is_what_the_control_group_grants():
    """host/ingest_mt.c: ma_cpu_budget() = the online CPUs cut by the control group's CPU quota (cgroup v2 cpu.max / v1 cfs_quota_us): the thread pools of the host side size
    themselves by it (a container that shows 256 CPUs and grants 16 throttles every thread of a process that runs 64)"""
    import math
    L = ma.lib()
    L.ma_cpu_budget.restype = C.c_int
    L.ma_ingest_threads.restype = C.c_int
    b = L.ma_cpu_budget()
    n = os.cpu_count()
    assert 1 <= b <= n
    want = n
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            want = min(want, math.ceil(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            want = min(want, math.ceil(q / per))
    except (OSError, ValueError):
        pass
    assert b <= want  # (the group's own file and its ancestors' may cut it further)
    if "MA_THREADS" not in os.environ:
        assert 1 <= L.ma_ingest_threads() <= min(b, 16)
