"""Pin the CPU oracle (oracle/ma_oracle.c) and the host-compiled device arithmetic (csrc/ma_core.h) against
the unmodified reference library built from /root/reference into oracle/_ref (CPU only, no GPU needed)."""
import ctypes as C
import os

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R
import stages as ST

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")

CASES = [
    ("lognormal", dict(reads=1500, lines=40000, seed=11, extra=[])),
    ("fixed", dict(reads=1200, lines=30000, seed=12, extra=["-L", "fixed"])),
    ("lowid", dict(reads=1000, lines=25000, seed=13, extra=["-i", "0.2"])),
    ("genome_order", dict(reads=1000, lines=25000, seed=14, extra=["-g"])),
    ("noisy", dict(reads=1500, lines=30000, seed=15, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"])),
]


@needs_ref
@pytest.mark.parametrize("name,cfg", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_stage_by_stage(name, cfg, tmpdir_s):
    paf = R.pafgen(os.path.join(tmpdir_s, "o_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    opt = ma.default_opt()
    ref = ST.ref_stages(paf, opt)
    ing = ma.Ingest(paf, opt)
    orc = ST.orc_stages(ing.hits, ing.n_seq, opt)
    # the sort is a total order here and an unstable one in the reference: compare as multisets, and the
    # graph only where the input has no (u,len) tie groups
    ST.compare(ref, orc, "ref vs oracle [%s]" % name, exact_order=False, graph=False)
    a, b = ref["sg_arcs"], orc["sg_arcs"]
    assert R.canon(a).tobytes() == R.canon(b).tobytes()
    keys = a["ul"]
    tie_free = len(np.unique(keys)) == len(keys)
    if tie_free:
        assert a.tobytes() == b.tobytes(), "sorted arc order differs on tie-free input"
        assert ref["tr_arcs"].tobytes() == orc["tr_arcs"].tobytes()
        assert ref["n_red"] == orc["n_red"]
    R.ref().asg_destroy(ref["g"])
    ing.close()


def test_core_arith_matches_oracle_random():
    """csrc/ma_core.h (device arithmetic compiled for the host) == the oracle's literal restatement, on random
    and adversarial inputs incl. values around 2^31 where signedness matters."""
    core = C.CDLL(os.path.join(ma.PKG, "lib", "libma_core_host.so"))
    O = R.orc()
    rng = np.random.default_rng(7)
    N = 200000
    hits = np.zeros(N, dtype=ma.HIT_DT)
    big = rng.random(N) < 0.05
    def coord(hi):
        x = rng.integers(0, hi, N, dtype=np.uint64)
        x[big] = rng.integers(2**31 - 50000, 2**31 + 50000, big.sum(), dtype=np.uint64)
        return x.astype(np.uint32)
    qs, span = coord(30000), rng.integers(0, 30000, N).astype(np.uint32)
    ts, tspan = coord(30000), rng.integers(0, 30000, N).astype(np.uint32)
    hits["qns"] = (rng.integers(0, 1000, N).astype(np.uint64) << 32) | qs
    hits["qe"] = qs + span
    hits["tn"] = rng.integers(0, 1000, N)
    hits["ts"], hits["te"] = ts, ts + tspan
    hits["mlrev"] = rng.integers(0, 2**31, N).astype(np.uint32) | (rng.integers(0, 2, N).astype(np.uint32) << 31)
    hits["bldel"] = rng.integers(0, 2**31, N)
    ql = rng.integers(0, 70000, N).astype(np.int32)
    tl = rng.integers(0, 70000, N).astype(np.int32)
    out4 = (C.c_uint32 * 4)()
    arc = np.zeros(1, dtype=ma.ARC_DT)
    core.core_hit2arc.argtypes = [C.c_uint32] * 6 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]
    n_arc = 0
    for i in range(0, N, 7):
        h = hits[i:i + 1]
        for (mh, fr, mo) in ((1000, .8, 2000), (1500, .5, 1000)):
            r0 = O.orc_hit2arc(h.ctypes.data, int(ql[i]), int(tl[i]), mh, fr, mo, arc.ctypes.data)
            r1 = core.core_hit2arc(int(h["qns"][0] >> 32), int(h["qns"][0] & 0xffffffff), int(h["qe"][0]), int(h["tn"][0]), int(h["ts"][0]),
                                   int(h["te"][0]), int(h["mlrev"][0] >> 31), int(ql[i]), int(tl[i]), mh, fr, mo, out4)
            assert r0 == r1, (i, r0, r1)
            if r0 >= 0:
                n_arc += 1
                assert int(arc["ul"][0]) == (out4[0] << 32 | out4[2]) and int(arc["v"][0]) == out4[1] and int(arc["oldel"][0]) == out4[3]
    assert n_arc > 100


@needs_ref
def test_cut_matches_reference_random():
    """mc_cut (device arithmetic) vs the reference's ma_hit_cut on random single-hit arrays"""
    core = C.CDLL(os.path.join(ma.PKG, "lib", "libma_core_host.so"))
    L = R.ref()
    rng = np.random.default_rng(3)
    N = 20000
    sub = np.zeros(64, dtype=ma.SUB_DT)
    s = rng.integers(0, 3000, 64).astype(np.uint32)
    sub["sdel"], sub["e"] = s, s + rng.integers(1000, 30000, 64).astype(np.uint32)
    c4 = (C.c_uint32 * 4)()
    kept = 0
    for i in range(N):
        h = np.zeros(1, dtype=ma.HIT_DT)
        q, t = int(rng.integers(0, 64)), int(rng.integers(0, 64))
        qs, ts = int(rng.integers(0, 20000)), int(rng.integers(0, 20000))
        span = int(rng.integers(1500, 12000))
        rev = int(rng.integers(0, 2))
        h["qns"], h["qe"], h["tn"], h["ts"], h["te"] = (q << 32) | qs, qs + span, t, ts, ts + span + int(rng.integers(-50, 50))
        h["mlrev"] = 500 | rev << 31
        c4[0], c4[1], c4[2], c4[3] = qs, int(h["qe"][0]), ts, int(h["te"][0])
        k1 = core.core_cut(c4, rev, int(sub["sdel"][q]), int(sub["e"][q]), int(sub["sdel"][t]), int(sub["e"][t]), 2000)
        k0 = L.ma_hit_cut(sub.ctypes.data, 2000, 1, h.ctypes.data)
        assert k0 == k1
        if k0:
            kept += 1
            assert (int(h["qns"][0]) & 0xffffffff, int(h["qe"][0]), int(h["ts"][0]), int(h["te"][0])) == (c4[0], c4[1], c4[2], c4[3])
    assert kept > 1000
