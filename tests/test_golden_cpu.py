"""CPU: the C oracle against the committed golden vectors (digests of the reference binary's own stage dumps,
tests/golden/golden.json).  The oracle's arrays are rendered with the reference's dump formats
(main.c:13-30, asm.c:41-55) by the few lines of formatting below, then digested like the golden files."""
import json
import os

import numpy as np

import miniasm_amd as ma
import refapi as R
import stages as ST

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")


def fmt_bed(names, sub):
    out = []
    for nm, s in zip(names, sub):
        ss, e = int(s["sdel"]) & 0x7FFFFFFF, int(s["e"])
        if ss != e:
            out.append("%s\t%d\t%d" % (nm, ss, e))
    return ("\n".join(out) + "\n").encode() if out else b""


def fmt_paf(names, sub, hits):
    out = []
    for h in hits:
        q, t = int(h["qns"]) >> 32, int(h["tn"])
        qs_, qe_ = int(sub[q]["sdel"]) & 0x7FFFFFFF, int(sub[q]["e"])
        ts_, te_ = int(sub[t]["sdel"]) & 0x7FFFFFFF, int(sub[t]["e"])
        out.append("%s:%d-%d\t%d\t%d\t%d\t%s\t%s:%d-%d\t%d\t%d\t%d\t%d\t%d\t255" % (
            names[q], qs_ + 1, qe_, qe_ - qs_, int(h["qns"]) & 0xFFFFFFFF, int(h["qe"]), "+-"[int(h["mlrev"]) >> 31],
            names[t], ts_ + 1, te_, te_ - ts_, int(h["ts"]), int(h["te"]), int(h["mlrev"]) & 0x7FFFFFFF, int(h["bldel"]) & 0x7FFFFFFF))
    return ("\n".join(out) + "\n").encode() if out else b""


def fmt_sg(names, sub, arcs):
    out = []
    for a in arcs:
        u, v = int(a["ul"]) >> 32, int(a["v"])
        q, t = u >> 1, v >> 1
        out.append("L\t%s:%d-%d\t%s\t%s:%d-%d\t%s\t%d:\tL1:i:%d" % (
            names[q], (int(sub[q]["sdel"]) & 0x7FFFFFFF) + 1, int(sub[q]["e"]), "+-"[u & 1],
            names[t], (int(sub[t]["sdel"]) & 0x7FFFFFFF) + 1, int(sub[t]["e"]), "+-"[v & 1], int(a["oldel"]) & 0x7FFFFFFF, int(a["ul"]) & 0xFFFFFFFF))
    return ("\n".join(out) + "\n").encode() if out else b""


def test_oracle_reproduces_golden_dumps(tmpdir_s):
    gold = json.load(open(GOLDEN))
    checked = 0
    for name, entry in gold["inputs"].items():
        cfg = entry["pafgen"]
        paf = R.pafgen(os.path.join(tmpdir_s, "gc_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
        assert R.digest(open(paf, "rb").read()) == entry["paf_digest"]
        opt = ma.default_opt()
        ing = ma.Ingest(paf, opt)
        S = ST.orc_stages(ing.hits, ing.n_seq, opt)
        names_all = ing.names()
        names = [nm for nm, m in zip(names_all, S["map"]) if m >= 0]
        assert len(names) == S["n_seq_new"]
        D = entry["dumps"]
        assert R.digest(fmt_bed(names, S["cont_sub"])) == D["-p bed"], name
        assert R.digest(fmt_paf(names, S["cont_sub"], S["cont"])) == D["-p paf"], name
        assert R.digest(fmt_paf(names_all, S["sub1"], S["cut1"])) == D["-p paf -S2"], name
        assert R.digest(fmt_paf(names_all, S["sub1"], S["flt"])) == D["-p paf -S3"], name
        assert R.digest(fmt_paf(names_all, S["subm"], S["cut2"])) == D["-p paf -S4"], name
        assert R.digest(fmt_sg(names, S["cont_sub"], S["sg_arcs"])) == D["-p sg -S5"], name
        assert R.digest(fmt_sg(names, S["cont_sub"], S["tr_arcs"])) == D["-p sg -S6"], name
        checked += 7
        ing.close()
    assert checked >= 28


def test_tiny_fixture_text_is_consistent_with_digests():
    """the human-readable tiny.* fixtures hash to what golden.json records for them"""
    here = os.path.dirname(GOLDEN)
    gold = json.load(open(GOLDEN))["inputs"]["tiny"]
    assert R.digest(open(os.path.join(here, "tiny.paf"), "rb").read()) == gold["paf_digest"]
    for key, want in gold["dumps"].items():
        fn = "tiny." + key.replace("-", "").replace(" ", "_").replace(",", "_") + ".txt"
        assert R.digest(open(os.path.join(here, fn), "rb").read()) == want, fn
