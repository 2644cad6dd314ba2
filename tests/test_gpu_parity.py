"""GPU parity tests proper: every HIP pass, called through the C ABI (include/mahip.h), against the C oracle
(bit-exact, order included: tie mode 0 = the stable total order the oracle computes) and, in the default tie mode,
against the unmodified reference library -- bit-exact INCLUDING the order of hits and arcs with equal sort keys on
every input (the synthetic inputs are full of equal (qid,qs) hit keys: every hit that covers the query's first base).
Run with `pytest -m gpu` on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
import stages as ST

pytestmark = pytest.mark.gpu

CASES = [
    ("lognormal", 3000, 80000, 41, []),
    ("fixed", 2500, 70000, 42, ["-L", "fixed"]),
    ("lowid", 2000, 50000, 43, ["-i", "0.2"]),
    ("genome_order", 2000, 50000, 44, ["-g"]),
    ("noisy", 4000, 90000, 45, ["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    ("deep_groups", 400, 300000, 46, []),                 # ~1500 hits per read: second tier of the coverage kernel
    ("deep_vertices", 1500, 1200000, 47, ["-L", "fixed"]),  # ~800 arcs per vertex: second tier of the reduction kernel
]


@pytest.mark.parametrize("name,reads,lines,seed,extra", CASES, ids=[c[0] for c in CASES])
def test_stages_match_oracle_and_reference(name, reads, lines, seed, extra, tmpdir_s, gpu_ctx):
    paf = R.pafgen(os.path.join(tmpdir_s, "g_%s.paf" % name), reads, lines, seed, extra)
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    gpu = ST.gpu_stages(gpu_ctx, ing.hits, ing.n_seq, opt)
    orc = ST.orc_stages(ing.hits, ing.n_seq, opt)
    ST.compare(orc, gpu, "oracle vs gpu [%s]" % name, exact_order=True, graph=True)
    assert orc["n_rem1"] == gpu["n_rem1"] and orc["n_rem2"] == gpu["n_rem2"]
    assert orc["map"].tobytes() == gpu["map"].tobytes()
    if "sg_idx" in gpu:
        idx = np.zeros(2 * gpu["n_seq_new"], dtype="<u8")
        R.orc().orc_arc_index(gpu["n_seq_new"], len(gpu["tr_arcs"]), gpu["tr_arcs"].ctypes.data, idx.ctypes.data)
        assert idx.tobytes() == gpu["tr_idx"].tobytes(), "CSR index differs"
    if R.have_ref():
        ref = ST.ref_stages(paf, opt)
        ST.compare(ref, gpu, "reference vs gpu, stable order [%s]" % name, exact_order=False, graph=False)
        assert R.canon(ref["sg_arcs"]).tobytes() == R.canon(gpu["sg_arcs"]).tobytes()
        # the default tie mode: every stage equals the reference's array, order included
        auto = ST.gpu_stages(gpu_ctx, ing.hits, ing.n_seq, opt, tie_mode=2)
        ST.compare(ref, auto, "reference vs gpu, default tie mode [%s]" % name, exact_order=True, graph=True)
        assert ref["tr_idx"].tobytes() == auto["tr_idx"].tobytes()
        keys = ref["sg_arcs"]["ul"]
        if len(np.unique(keys)) == len(keys):  # no (u,len) ties: nothing to repair, and the census must say so
            assert auto["tie"]["arc_tie_groups"] == 0 and auto["tie"]["arc_walk"] == 0
            assert ref["sg_arcs"].tobytes() == gpu["sg_arcs"].tobytes() and ref["tr_arcs"].tobytes() == gpu["tr_arcs"].tobytes()
        else:
            assert auto["tie"]["arc_tie_groups"] > 0 and auto["tie"]["arc_walk"] == 1
        R.ref().asg_destroy(ref["g"])
    ing.close()


def test_sort_random_keys_and_ties(gpu_ctx):
    """radix sort on adversarial keys: heavy ties, wide query ids, zero starts, sizes around tile boundaries"""
    rng = np.random.default_rng(9)
    for n, nq, ns in ((1, 1, 1), (2, 1, 1), (63, 5, 3), (2048, 7, 2), (2049, 300, 50), (100000, 70000, 60000), (300001, 17, 5), (50000, 1 << 24, 1 << 20),
                      (5000, 100, 100), (200000, 16000, 16000), (70000, 2000000, 1000)):  # digits of exactly 7 bits: the scatter's compile-time width
        h = np.zeros(n, dtype=ma.HIT_DT)
        h["qns"] = (rng.integers(0, nq, n).astype(np.uint64) << 32) | rng.integers(0, ns, n).astype(np.uint64)
        h["qe"] = np.arange(n)  # distinguishes tied records: stability is observable
        h["tn"] = rng.integers(0, nq, n)
        gpu_ctx.set_exact_ties(0)  # the stable total order (key, input position)
        gpu_ctx.hits_upload(h, int(nq))
        gpu_ctx.sort()
        got = gpu_ctx.hits_download()
        exp = h.copy()
        R.orc().orc_hit_sort(n, exp.ctypes.data)
        assert got.tobytes() == exp.tobytes(), "sort differs for n=%d" % n
        gpu_ctx.set_exact_ties(2)  # default: ties as the reference's in-place radix sort leaves them (hit.c:19-22)
        if R.have_ref():
            gpu_ctx.hits_upload(h, int(nq))
            gpu_ctx.sort()
            got = gpu_ctx.hits_download()
            LR = R.ref()
            LR.radix_sort_hit.argtypes = [C.c_void_p, C.c_void_p]
            LR.radix_sort_hit.restype = None
            exp = h.copy()
            LR.radix_sort_hit(exp.ctypes.data, exp.ctypes.data + n * 32)
            assert got.tobytes() == exp.tobytes(), "reference tie order differs for n=%d" % n


def test_runs_sort_takes_no_ids_from_outside_the_dictionary(gpu_ctx):
    """the sort on RUNS of records packs query id | position | length into one word: an id >= n_seq (a caller's contract violation) would lose its high bits there and
    pass for another read's.  k_hit_keys_runs sees it and the sort falls back to records, where the id is reported or carried in full; and the helper says
    which path a sort took"""
    rng = np.random.default_rng(12)
    n, nq = 40000, 3000
    q = np.repeat(rng.permutation(nq)[: n // 40], 40).astype(np.uint64)  # runs of 40 records: the runs path is worth it
    h = np.zeros(n, dtype=ma.HIT_DT)
    h["qns"] = (q << 32) | rng.integers(0, 5000, n).astype(np.uint64)
    h["qe"] = np.arange(n)
    h["tn"] = rng.integers(0, nq, n)
    gpu_ctx.set_exact_ties(0)
    gpu_ctx.hits_upload(h, nq)
    gpu_ctx.set_run_stride(1)
    gpu_ctx.sort()
    assert gpu_ctx.sorted_runs() > 0, "runs of 40 records under stride 1: the runs path was expected"
    exp = h.copy()
    R.orc().orc_hit_sort(n, exp.ctypes.data)
    assert gpu_ctx.hits_download().tobytes() == exp.tobytes()
    bad = h.copy()
    bad["qns"][1234] = (np.uint64(nq + 4096 + 7) << np.uint64(32)) | np.uint64(5)  # 12 id bits: 7191 would read as 3095 & 4095
    gpu_ctx.hits_upload(bad, nq)
    gpu_ctx.set_run_stride(1)
    try:
        gpu_ctx.sort()
    except Exception:
        pass  # the record path may refuse such an input outright
    assert gpu_ctx.sorted_runs() == 0, "an id outside the dictionary went through the runs path"
    gpu_ctx.set_exact_ties(2)


def test_sort_with_keys_wider_than_64_bits(gpu_ctx):
    """query id, query start and input position together need 65..67 bits: the packed key drops the low position bits and the gather
    picks the record among the 2^drop neighbours (rank inside the run of equal keys); beyond that the (key, value) pair sort takes over"""
    rng = np.random.default_rng(10)
    LR = R.ref() if R.have_ref() else None
    for n, qvals, svals in ((100000, [0, 5, (1 << 24) - 1], [0, 7, (1 << 24) - 1]),            # 17 + 24 + 24 = 65 bits, three distinct values each: huge tie runs
                            (300001, None, None),                                              # 19 + 24 + 23 = 66 bits, random keys
                            (300001, [1, (1 << 24) - 2], [3, (1 << 23) + 1]),                  # 66 bits, tie runs of ~75000
                            (70000, [0, (1 << 26) - 1], None)):                                # 17 + 26 + 24 = 67 bits
        h = np.zeros(n, dtype=ma.HIT_DT)
        q = rng.choice(np.array(qvals, dtype=np.uint64), n) if qvals else rng.integers(0, 1 << 24, n).astype(np.uint64)
        s = rng.choice(np.array(svals, dtype=np.uint64), n) if svals else rng.integers(0, 1 << (24 if n == 70000 else 23), n).astype(np.uint64)
        if not svals:
            s[0] = (1 << (24 if n == 70000 else 23)) - 1
        if not qvals:
            q[0] = (1 << 24) - 1
        h["qns"] = (q << np.uint64(32)) | s
        h["qe"] = np.arange(n)
        h["tn"] = rng.integers(0, 1000, n)
        nq = int(q.max()) + 1
        for mode in (0, 2):
            gpu_ctx.set_exact_ties(mode)
            gpu_ctx.hits_upload(h, nq)
            gpu_ctx.sort()
            got = gpu_ctx.hits_download()
            exp = h.copy()
            if mode == 0:
                R.orc().orc_hit_sort(n, exp.ctypes.data)
            elif LR is not None:
                LR.radix_sort_hit.argtypes = [C.c_void_p, C.c_void_p]
                LR.radix_sort_hit.restype = None
                LR.radix_sort_hit(exp.ctypes.data, exp.ctypes.data + n * 32)
            else:
                continue
            assert got.tobytes() == exp.tobytes(), "wide-key sort differs (n=%d, tie mode %d)" % (n, mode)
    gpu_ctx.set_exact_ties(2)


def test_empty_and_degenerate_inputs(gpu_ctx):
    opt = ma.default_opt()
    # no hits at all
    S = ST.gpu_stages(gpu_ctx, np.zeros(0, dtype=ma.HIT_DT), 5, opt)
    assert len(S["sorted"]) == 0 and S["n_seq_new"] == 0 and len(S["tr_arcs"]) == 0
    assert S["sub1"].tobytes() == np.zeros(5, ma.SUB_DT).tobytes()
    # only self hits, reads that never appear as a query, a read with a single hit
    h = np.zeros(4, dtype=ma.HIT_DT)
    h["qns"] = [(0 << 32) | 0, (0 << 32) | 10, (2 << 32) | 0, (3 << 32) | 100]
    h["qe"] = [5000, 6000, 4000, 4100]
    h["tn"] = [0, 0, 2, 1]
    h["ts"], h["te"] = [0, 10, 0, 0], [5000, 6000, 4000, 4000]
    h["mlrev"], h["bldel"] = [900, 900, 900, 900], [5000, 5990, 4000, 4000]
    G = ST.gpu_stages(gpu_ctx, h, 6, opt)
    O = ST.orc_stages(h, 6, opt)
    ST.compare(O, G, "degenerate", exact_order=True)


def test_flags_and_ragged_groups_vs_oracle(gpu_ctx, tmpdir_s):
    """ragged group sizes (1 .. >64 .. >2048 events) in one input, -b style input (no mirrored hits)"""
    paf = R.pafgen(os.path.join(tmpdir_s, "rag.paf"), 900, 120000, 51, ["-S", "0.9"])
    opt = ma.default_opt()
    for bi_dir in (True, False):
        ing = ma.Ingest(paf, opt, bi_dir=bi_dir)
        sizes = np.bincount((ing.hits["qns"] >> 32).astype(np.int64))
        assert sizes.max() > 64
        G = ST.gpu_stages(gpu_ctx, ing.hits, ing.n_seq, opt)
        O = ST.orc_stages(ing.hits, ing.n_seq, opt)
        ST.compare(O, G, "ragged bi_dir=%s" % bi_dir, exact_order=True)
        ing.close()


def test_custom_thresholds_vs_oracle(gpu_ctx, tmpdir_s):
    paf = R.pafgen(os.path.join(tmpdir_s, "thr.paf"), 2000, 50000, 52, ["-L", "uniform", "-d", "0.2", "-x", "0.05", "-i", "0.1"])
    for (dp, iden, span, hang, frac, fuzz) in ((2, .05, 1500, 500, .7, 500), (5, .1, 2500, 2000, .9, 0), (3, .2, 2000, 1000, .8, 3000)):
        opt = ma.default_opt()
        opt.min_dp, opt.min_iden, opt.min_span, opt.max_hang, opt.int_frac, opt.gap_fuzz = dp, iden, span, hang, frac, fuzz
        opt.min_ovlp = span
        ing = ma.Ingest(paf, opt)
        G = ST.gpu_stages(gpu_ctx, ing.hits, ing.n_seq, opt)
        O = ST.orc_stages(ing.hits, ing.n_seq, opt)
        ST.compare(O, G, "thresholds %r" % ((dp, iden, span, hang, frac, fuzz),), exact_order=True)
        ing.close()


def test_graph_passes_on_uploaded_graph(gpu_ctx, tmpdir_s):
    """per-symbol graph path: upload a host graph, del_short / del_trans / symm, compare with the oracle"""
    paf = R.pafgen(os.path.join(tmpdir_s, "gr.paf"), 2500, 60000, 53, ["-L", "fixed", "-d", "0.3", "-x", "0.04"])
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    O = ST.orc_stages(ing.hits, ing.n_seq, opt)
    ns, arcs = O["n_seq_new"], O["sg_arcs"].copy()
    seq = O["sg_seq"].astype("<u4").copy()
    idx = np.zeros(2 * ns, dtype="<u8")
    R.orc().orc_arc_index(ns, len(arcs), arcs.ctypes.data, idx.ctypes.data)
    g = ma.Asg()
    g.arc, g.n_arc_srt, g.m_arc = arcs.ctypes.data, len(arcs) | 1 << 31, len(arcs)
    g.seq, g.n_seq_symm, g.m_seq = seq.ctypes.data, ns, ns
    g.idx = idx.ctypes.data
    ma._chk(ma.lib().mahip_asg_upload(gpu_ctx.h, C.byref(g)), "asg_upload")
    a0, s0, i0 = gpu_ctx.asg_download()
    assert a0.tobytes() == arcs.tobytes() and s0.tobytes() == seq.tobytes() and i0.tobytes() == idx.tobytes()
    for ratio in (.5, .7, .9):
        exp = arcs.copy()
        n0 = R.orc().orc_arc_del_short(ns, len(exp), exp.ctypes.data, idx.ctypes.data, ratio)
        ma._chk(ma.lib().mahip_asg_upload(gpu_ctx.h, C.byref(g)), "asg_upload")
        n1 = gpu_ctx.del_short(ratio)
        assert n0 == n1
        sdel = (seq >> 31).astype(np.uint8)
        m = R.orc().orc_arc_rm(len(exp), exp.ctypes.data, sdel.ctypes.data)
        got, _, _ = gpu_ctx.asg_download()
        assert got.tobytes() == exp[:m].tobytes()
    ing.close()


def random_hits(seed):
    """hit arrays no overlapper writes: a few very deep reads, coordinates on a coarse grid (ties everywhere, zero-length and full-length overlaps),
    start > end, self hits, ml > bl, bl = 0 -- the passes are integer arithmetic with C's wrap-around rules, so garbage in must give the oracle's
    garbage out, bit for bit"""
    rng = np.random.default_rng(seed)
    R_ = int(rng.choice([1, 3, 20, 200, 2000]))
    n = int(rng.choice([1, 2, 50, 3000, 60000]))
    mode = int(rng.integers(0, 4))
    rl = rng.integers(500, 20000, R_)
    q = rng.integers(0, R_, n) if mode != 3 else np.minimum(rng.geometric(0.05, n) - 1, R_ - 1)
    t = rng.integers(0, R_, n)
    ql, tl = rl[q], rl[t]
    if mode == 0:
        qs = (rng.random(n) * ql * 0.8).astype(np.int64); qe = qs + 1 + (rng.random(n) * (ql - qs - 1)).astype(np.int64)
        ts = (rng.random(n) * tl * 0.8).astype(np.int64); te = ts + 1 + (rng.random(n) * (tl - ts - 1)).astype(np.int64)
    elif mode == 1:
        g = 250
        qs = rng.integers(0, 8, n) * g; qe = np.minimum(qs + rng.integers(0, 40, n) * g, ql)
        ts = rng.integers(0, 8, n) * g; te = np.minimum(ts + rng.integers(0, 40, n) * g, tl)
    else:
        qs = rng.integers(0, ql + 1); qe = rng.integers(0, ql + 1)
        ts = rng.integers(0, tl + 1); te = rng.integers(0, tl + 1)
    bl = rng.integers(0, 30000, n)
    ml = (bl * rng.random(n) * 1.2).astype(np.int64)
    h = np.zeros(n, dtype=ma.HIT_DT)
    h["qns"] = (q.astype(np.uint64) << np.uint64(32)) | (qs.astype(np.uint64) & np.uint64(0xffffffff))
    h["qe"] = qe.astype(np.uint32); h["tn"] = t.astype(np.uint32); h["ts"] = ts.astype(np.uint32); h["te"] = te.astype(np.uint32)
    h["mlrev"] = (ml.astype(np.uint32) & np.uint32(0x7fffffff)) | (rng.integers(0, 2, n).astype(np.uint32) << np.uint32(31))
    h["bldel"] = bl.astype(np.uint32) & np.uint32(0x7fffffff)
    return h, R_


@pytest.mark.parametrize("block", range(4))
def test_random_hit_arrays_match_the_oracle_at_every_stage(block, gpu_ctx):
    opt = ma.default_opt()
    for seed in range(block * 8, block * 8 + 8):
        h, n_seq = random_hits(seed)
        orc = ST.orc_stages(h, n_seq, opt)
        gpu = ST.gpu_stages(gpu_ctx, h, n_seq, opt)
        ST.compare(orc, gpu, "random hits, seed %d" % seed, exact_order=True, graph=True)


@pytest.mark.parametrize("n_seq,n,where", [(2048, 1, "last"), (2049, 100, "first"), (6000, 300, "middle"), (6000, 5000, "ends"), (70000, 20000, "sparse"), (70000, 3, "last")])
def test_group_offsets_when_most_reads_have_no_hits(n_seq, n, where, gpu_ctx):
    """the group offsets come out of the sort's last pass (radix.hip: RsGroups) and the reads without hits are closed afterwards, tile of 2048 ids by tile:
    ids in use only at one end, in one tile of several, or thinly spread -- every stage behind the sort sees wrong groups if one offset is off"""
    rng = np.random.default_rng(n_seq + n)
    if where == "last":
        q = np.full(n, n_seq - 1)
    elif where == "first":
        q = np.zeros(n, dtype=np.int64)
    elif where == "middle":
        q = rng.integers(2500, 2600, n)
    elif where == "ends":
        q = np.where(rng.integers(0, 2, n) == 0, rng.integers(0, 3, n), n_seq - 1 - rng.integers(0, 3, n))
    else:
        q = rng.choice(n_seq, 40, replace=False)[rng.integers(0, 40, n)]
    t = rng.integers(0, n_seq, n)
    qs = rng.integers(0, 5000, n); qe = qs + rng.integers(1, 5000, n)
    ts = rng.integers(0, 5000, n); te = ts + rng.integers(1, 5000, n)
    bl = rng.integers(1, 6000, n)
    h = np.zeros(n, dtype=ma.HIT_DT)
    h["qns"] = (q.astype(np.uint64) << np.uint64(32)) | qs.astype(np.uint64)
    h["qe"] = qe.astype(np.uint32); h["tn"] = t.astype(np.uint32); h["ts"] = ts.astype(np.uint32); h["te"] = te.astype(np.uint32)
    h["mlrev"] = (bl * 9 // 10).astype(np.uint32); h["bldel"] = bl.astype(np.uint32)
    opt = ma.default_opt()
    opt.min_dp = 1
    orc = ST.orc_stages(h, n_seq, opt)
    gpu = ST.gpu_stages(gpu_ctx, h, n_seq, opt)
    ST.compare(orc, gpu, "sparse ids: %d reads, %d hits, %s" % (n_seq, n, where), exact_order=True, graph=True)
