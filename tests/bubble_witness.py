"""Hand-made graphs on which asg_bub_backtrack (reference asg.c:338-357) sets seq.del = 0 for a read that is DEAD when the pop happens -- the one thing the
cleaners' fixpoint cannot write as a stamp (miniasm_amd/csrc/clean_core.h, ASSUMPTION + CLAIM).  clean_core.h proves that a symmetric, clean graph never
gets there; these two are outside that contract and the reference still answers on them:

  unclean     a plain bubble 0+ -> {1+, 2+} -> 3+ (-> 4+) whose sink read 3 carries seq.del = 1 although its arcs are live (a graph nobody ran asg_cleanup on)
  asymmetric  is_symm is set but mirror arcs are missing.  Pop A (source 0+) trims the tip branch 0+ -> W+ -> U+ and keeps 0+ -> P+; W dies, but
              W- -> Z- survives because A looks for the mirror W- -> 0- of the arc it walked and there is none.  Pop B (source S2+, a LATER vertex of
              the same sweep) walks S2+ -> W+ -- an arc A never saw, it has no mirror either --, W+ still "expects" one incoming arc (count_out(W-) = 1),
              ends as B's sink and is brought back to life: deleted by an earlier pop, resurrected by a later one.

Used by tests/test_clean_core_cpu.py (host harness) and tests/test_gpu_graph_api.py (per-symbol asg_pop_bubble on the device)."""
import numpy as np

import miniasm_amd as ma
import refapi as R


def _pack(n_seq, rows, dead=()):
    a = np.zeros(len(rows), dtype=ma.ARC_DT)
    for i, (u, v, ln, ol) in enumerate(rows):
        a[i] = ((u << 32) | ln, v, ol)
    a = a[np.argsort(a["ul"], kind="stable")]
    seq = np.full(n_seq, 9000, dtype="<u4")
    for r in dead:
        seq[r] |= 1 << 31
    idx = np.zeros(2 * n_seq, dtype="<u8")
    R.orc().orc_arc_index(n_seq, len(a), a.ctypes.data, idx.ctypes.data)
    return a, seq, idx


def _both(u, v, ln, ol=4000):
    return [(u, v, ln, ol), (v ^ 1, u ^ 1, ln + 7, ol)]


def unclean():
    rows = _both(0, 2, 500) + _both(0, 4, 600) + _both(2, 6, 700) + _both(4, 6, 650) + _both(6, 8, 400)  # (the sink needs an arc of its own, or it counts as a tip)
    return _pack(5, rows, dead=(3,)), "sink read 3 is flagged deleted but has live arcs"


def asymmetric():
    S1, P, W, U, S2, Q, U2, Z, V = (2 * r for r in range(9))  # '+' vertices; x ^ 1 = the '-' strand
    rows = []
    rows += _both(S1, P, 100)            # kept branch of pop A
    rows += [(S1, W, 200, 4000)]         # tip branch; NO mirror W- -> S1-
    rows += _both(W, U, 100)             # U+ has no arcs: a tip
    rows += _both(P, V, 100)             # P+ needs an arc of its own to be a sink rather than a tip
    rows += [(W ^ 1, Z ^ 1, 100, 4000)]  # the one arc out of W-; NO mirror Z+ -> W+
    rows += [(S2, W, 100, 4000)]         # pop B's way into W+; NO mirror
    rows += _both(S2, Q, 200)
    rows += _both(Q, U2, 100)
    return _pack(9, rows), "read 2 (W) is deleted by the pop from vertex 0 and revived by the pop from vertex 8"


WITNESSES = {"unclean": unclean, "asymmetric": asymmetric}
