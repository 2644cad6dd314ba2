"""The claims the automatic tie-order repair rests on, checked on the CPU against the unmodified
reference library:

 (1) the order the reference leaves the arcs in after ma_sg_gen is reproduced by its sort procedure (host/refsort.c)
     applied to the arcs IN THE REFERENCE'S PUSH ORDER with keys built from the SQUEEZED read ids -- including inputs on
     which > 96 % of the reads are dropped by containment (with unsqueezed ids the bucket sizes, hence the insertion-sort
     cut-off of ksort.h:182, differ and the order comes out wrong: ADVICE round 1);
 (2) a stable sort of the hits differs from the reference's only inside runs of equal (qid,qs), so the reference's push
     order is the stable push order re-ranked inside such runs.

The device side of the repair (census, slot bookkeeping, re-gather) is covered by tests/test_gpu_cli.py."""
import ctypes as C
import os

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
import stages as ST

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")

CASES = {
    "grid16": dict(reads=3000, lines=80000, seed=5, extra=["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"]),
    "grid400_lognormal": dict(reads=4000, lines=120000, seed=7, extra=["-q", "400", "-d", "0.2"]),
    "grid50_fixed": dict(reads=2500, lines=70000, seed=8, extra=["-q", "50", "-L", "fixed", "-d", "0.1"]),
    "grid400_deep": dict(reads=30000, lines=1500000, seed=9, extra=["-q", "400", "-d", "0.2", "-x", "0.03"]),  # deep coverage: 97 % of the reads are contained
}


def _refsort_perm(keys):
    L = ma.lib()
    L.ma_refsort_perm.restype = C.c_int
    L.ma_refsort_perm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    keys = np.ascontiguousarray(keys, dtype="<u8")
    perm = np.zeros(len(keys), dtype=np.uint32)
    assert L.ma_refsort_perm(keys.ctypes.data, len(keys), perm.ctypes.data) == 0
    return perm


@needs_ref
@pytest.mark.parametrize("name", list(CASES))
def test_arc_order_from_squeezed_keys_in_push_order(name, tmpdir_s):
    cfg = CASES[name]
    paf = R.pafgen(os.path.join(tmpdir_s, "tr_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    opt = ma.default_opt()
    S = ST.ref_stages(paf, opt)
    R.ref().asg_destroy(S["g"])
    hits, ns = S["cont"], S["n_seq_new"]  # the reference's hits after ma_hit_contained: its order, squeezed ids
    O = R.orc()
    arcs = np.zeros(max(len(hits), 1), dtype=ma.ARC_DT)
    slen = np.zeros(max(ns, 1), dtype="<u4")
    sdel = np.zeros(max(ns, 1), dtype=np.uint8)
    O.orc_sg_candidates.restype = C.c_size_t
    O.orc_sg_candidates.argtypes = [C.POINTER(ma.MaOpt), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    m = O.orc_sg_candidates(C.byref(opt), ns, S["cont_sub"].ctypes.data if ns else None, None, None, len(hits), hits.ctypes.data,
                            arcs.ctypes.data, slen.ctypes.data, sdel.ctypes.data)
    arcs = arcs[:m]
    u, v = (arcs["ul"] >> np.uint64(33)).astype(np.int64), (arcs["v"] >> 1).astype(np.int64)
    push = arcs[(sdel[u] == 0) & (sdel[v] == 0)]  # asg_arc_rm keeps the push order (asg.c:57-70)
    want = S["sg_arcs"]
    assert len(push) == len(want)
    n_groups = int((np.unique(want["ul"], return_counts=True)[1] > 1).sum())
    assert n_groups >= 5, "input is supposed to be tie-rich"
    if name == "grid400_deep":
        assert ns < 0.04 * S["n_seq"], "this case is about inputs where > 96 %% of the reads are dropped (%d of %d left)" % (ns, S["n_seq"])
    got = push[_refsort_perm(push["ul"])]
    assert got.tobytes() == want.tobytes(), "%s: the host walk over squeezed keys does not give the reference's arc order" % name


@needs_ref
def test_stable_hit_order_differs_only_inside_tie_runs(tmpdir_s):
    cfg = CASES["grid16"]
    paf = R.pafgen(os.path.join(tmpdir_s, "tr_hits.paf"), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    hits = ing.hits.copy()
    ing.close()
    hits["bldel"] &= 0x7FFFFFFF
    perm = _refsort_perm(hits["qns"])                     # reference order: position i holds input record perm[i]
    sidx = np.argsort(hits["qns"], kind="stable")          # stable order: slot s holds input record sidx[s]
    inv = np.empty(len(hits), dtype=np.int64)
    inv[perm] = np.arange(len(hits))
    hrank = inv[sidx]                                      # what hits_reference_rank() computes on the device
    moved = np.nonzero(hrank != np.arange(len(hits)))[0]
    assert len(moved) > 0, "input is supposed to have tied hits"
    key = hits["qns"][sidx]
    assert (key[moved] == key[hrank[moved]]).all()         # a slot only ever moves inside its run of equal keys
    ref = hits[perm]
    LR = R.ref()
    LR.radix_sort_hit.argtypes = [C.c_void_p, C.c_void_p]
    LR.radix_sort_hit.restype = None
    chk = hits.copy()
    LR.radix_sort_hit(chk.ctypes.data, chk.ctypes.data + len(chk) * 32)
    assert chk.tobytes() == ref.tobytes()
    out = np.empty_like(ref)
    out[hrank] = hits[sidx]                                # export through hrank = the reference's array
    assert out.tobytes() == ref.tobytes()


def _packed(keys, bl, bi, shift_top):
    """the elements radix.hip: k_pack_keys sends down: (hi << bl | lo) << bi | input position; with shift_top >= 32 hi goes without its bits from shift_top - 32 up and
    the key's digit at shift_top travels in a byte array of its own (16 zero bytes behind it)"""
    hi, lo = keys >> np.uint64(32), keys & np.uint64(0xffffffff)
    dig = None
    if shift_top >= 0:
        dig = np.zeros(len(keys) + 16, dtype=np.uint8)
        dig[:len(keys)] = ((keys >> np.uint64(shift_top)) & np.uint64(0xff)).astype(np.uint8)
        hi = hi & np.uint64((1 << (shift_top - 32)) - 1)
    pk = ((hi << np.uint64(bl) | lo) << np.uint64(bi)) | np.arange(len(keys), dtype=np.uint64)
    return np.ascontiguousarray(pk), dig


@pytest.mark.parametrize("n,n_ids,top_apart,threads", [(300000, 200000, False, "4"), (300000, 200000, True, "4"), (40000, 3000, True, "1"), (500000, 70000, True, "8"), (200000, 100, False, "3")])
def test_restricted_walk_gives_the_wanted_reads_their_order(n, n_ids, top_apart, threads, monkeypatch):
    """ma_refsort_packed_wanted (the tie walk restricted to the reads whose conflicts the arc sort can see) leaves, at the positions of a WANTED read's hits, exactly what
    the whole sort leaves there -- for every subset of wanted reads, with the words carrying the whole key or not its top digit, on one thread and on several"""
    monkeypatch.setenv("MA_THREADS", threads)
    L = ma.lib()
    L.ma_refsort_packed.restype = C.c_int
    L.ma_refsort_packed.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.ma_refsort_packed_wanted.restype = C.c_int
    L.ma_refsort_packed_wanted.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(n + n_ids)
    ids = rng.integers(0, n_ids, n, dtype=np.uint64)
    ids[: n // 3] = np.sort(ids[: n // 3])  # a PAF lists a query's lines together: stretches of one id
    qs = rng.choice(np.array([0, 0, 0, 5, 17, 300, 301, 4000, 16383], dtype=np.uint64), n)  # equal keys everywhere
    keys = ids << np.uint64(32) | qs
    bh, bl, bi = int(n_ids - 1).bit_length(), 14, int(n - 1).bit_length()
    shift_top = ((32 + bh - 1) & ~7) if top_apart else -1
    if top_apart and shift_top < 32:
        pytest.skip("ids of one byte: no digit to take apart")
    full, dig = _packed(keys, bl, bi, shift_top)
    dptr = dig.ctypes.data if dig is not None else None
    assert L.ma_refsort_packed(full.ctypes.data, n, bl, bi, shift_top, dptr) == 0
    order = np.argsort(keys, kind="stable")
    start = np.searchsorted(keys[order], np.arange(n_ids + 1, dtype=np.uint64) << np.uint64(32))  # where a read's hits stand in ANY sorted order
    for frac in (0.0005, 0.02, 0.5):
        want = rng.random(n_ids) < frac
        want[int(ids[0])] = True
        wcum = np.zeros(n_ids + 1, dtype=np.uint32)
        wcum[1:] = np.cumsum(want)
        part, _ = _packed(keys, bl, bi, shift_top)
        assert L.ma_refsort_packed_wanted(part.ctypes.data, n, bl, bi, shift_top, dptr, wcum.ctypes.data, n_ids) == 0
        sel = np.zeros(n, dtype=bool)
        for r in np.flatnonzero(want):
            sel[start[r]:start[r + 1]] = True
        assert sel.any() and (part[sel] == full[sel]).all(), frac
        assert (np.sort(part) == np.sort(full)).all()  # still a permutation of the input
        if frac < 0.001:
            assert (part != full).any()  # (and the work was left out: what nobody asked for is not in order)
