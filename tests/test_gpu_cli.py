"""End-to-end parity on the GPU: the miniasm CLI (resident pipeline) and the link-level drop-in (the
reference's own main.o linked against libminiasm_amd.so, per-symbol ABI) against the unmodified reference
binary, for every dump the reference can produce (-p bed|paf|sg|ug x -S stage), plus committed golden digests."""
import json
import os

import pytest

import miniasm_amd as ma
import refapi as R

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")

DUMPS = [["-p", "bed"], ["-p", "paf", "-S2"], ["-p", "paf", "-S3"], ["-p", "paf", "-S4"], ["-p", "paf"],
         ["-p", "sg", "-S5"], ["-p", "sg", "-S6"], ["-p", "sg", "-S7"], ["-p", "sg", "-S9"], ["-p", "sg", "-S10"], ["-p", "sg"], ["-p", "ug"]]

INPUTS = {  # name -> pafgen arguments (all arc-tie-free: checked when the golden file was made)
    "lognormal": dict(reads=3000, lines=80000, seed=41, extra=[]),
    "fixed": dict(reads=2500, lines=70000, seed=42, extra=["-L", "fixed"]),
    "noisy": dict(reads=4000, lines=90000, seed=63, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    "longnames": dict(reads=3000, lines=80000, seed=41, extra=["-N", "m64011_190830_220126/"]),  # names as real PacBio / ONT files carry them (the dictionary compares text)
}

EXTRA_ARGS = [["-1"], ["-2", "-p", "sg"], ["-b"], ["-R"], ["-R", "-p", "paf"], ["-R", "-h", "4000", "-p", "bed"], ["-c", "2", "-s", "1500", "-h", "500", "-I", "0.7", "-g", "500", "-e", "3", "-d", "30000"],
              ["-n", "4", "-r", "0.8,0.4", "-F", "0.9"], ["-o", "1000", "-m", "200", "-i", "0.1"], ["-1", "-2", "-p", "sg"],
              ["-b", "-S", "5", "-p", "ug"], ["-b", "-S", "4", "-p", "ug"]]  # unitigs of a graph that is NOT symmetric (ug.hip's one-lane sweep)


def _gen(tmpdir_s, name):
    cfg = INPUTS[name]
    return R.pafgen(os.path.join(tmpdir_s, "cli_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])


def _same(binary, args, paf, ref_out, ref_log, what):
    out, log = R.run_cli(binary, args, paf)
    assert R.norm_lines(out) == R.norm_lines(ref_out), "%s %s: output differs from the reference" % (what, " ".join(args))
    mine, theirs = R.counters(log), R.counters(ref_log)
    mine = [x for x in mine if not x.startswith("main: Version")]
    assert mine == theirs, "%s %s: log counters differ\n%s\nvs\n%s" % (what, " ".join(args), "\n".join(mine), "\n".join(theirs))


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", list(INPUTS))
def test_cli_and_dropin_match_reference_binary(name, tmpdir_s):
    paf = _gen(tmpdir_s, name)
    for args in DUMPS:
        ref_out, ref_log = R.run_cli(R.REF_BIN, args, paf)
        _same(ma.CLI_PATH, args, paf, ref_out, ref_log, "cli[%s]" % name)
    for args in (["-p", "bed"], ["-p", "paf"], ["-p", "sg", "-S5"], ["-p", "sg", "-S6"], ["-p", "ug"]):
        ref_out, ref_log = R.run_cli(R.REF_BIN, args, paf)
        _same(R.DROPIN_BIN, args, paf, ref_out, ref_log, "dropin[%s]" % name)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_cli_options_match_reference_binary(tmpdir_s, monkeypatch):
    paf = _gen(tmpdir_s, "noisy")
    for args in EXTRA_ARGS:
        ref_out, ref_log = R.run_cli(R.REF_BIN, args, paf)
        _same(ma.CLI_PATH, args, paf, ref_out, ref_log, "cli-opts")
    # -R (ma_hit_no_cont, hit.c:38-68): through the device parser (default) and through the host reader; on an input where the
    # pre-filter really drops reads (length spread: short reads clearly inside long ones)
    paf2 = R.pafgen(os.path.join(tmpdir_s, "cli_R.paf"), 3000, 90000, 64, ["-d", "0.1"])
    for args in (["-R"], ["-R", "-p", "paf", "-S2"], ["-R", "-p", "bed"]):
        ref_out, ref_log = R.run_cli(R.REF_BIN, args, paf2)
        assert "dropped 0 contained reads" not in ref_log, "this input is supposed to trigger the -R pre-filter"
        for host in (False, True):
            if host:
                monkeypatch.setenv("MA_HOST_PARSE", "1")
            else:
                monkeypatch.delenv("MA_HOST_PARSE", raising=False)
            _same(ma.CLI_PATH, args, paf2, ref_out, ref_log, "cli -R (%s parser)" % ("host" if host else "device"))
        monkeypatch.delenv("MA_HOST_PARSE", raising=False)


def test_cli_matches_golden_digests(tmpdir_s):
    """digests of the reference's normalised dumps, committed under tests/golden (made by tests/golden/make_golden.py)"""
    gold = json.load(open(GOLDEN))
    n = 0
    for name, entry in gold["inputs"].items():
        cfg = entry["pafgen"]
        paf = R.pafgen(os.path.join(tmpdir_s, "gold_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
        assert R.digest(open(paf, "rb").read()) == entry["paf_digest"], "pafgen is not reproducing the golden input"
        for key, want in entry["dumps"].items():
            out, _ = R.run_cli(ma.CLI_PATH, key.split(), paf)
            assert R.digest(out) == want, "golden mismatch: %s %s" % (name, key)
            n += 1
    assert n >= 10


def test_cli_gz_and_stdin(tmpdir_s):
    import gzip
    import subprocess
    paf = _gen(tmpdir_s, "lognormal")
    base, _ = R.run_cli(ma.CLI_PATH, [], paf)
    gz = paf + ".gz"
    with open(paf, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    out, _ = R.run_cli(ma.CLI_PATH, [], gz)
    assert out == base
    r = subprocess.run([ma.CLI_PATH, "-"], stdin=open(paf, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == base


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_cli_unitig_sequences_with_reads_file(tmpdir_s):
    """-f reads.fa: unitig sequences stitched from the reads (reference main.c:193, asm.c:216-290)"""
    import random
    paf = _gen(tmpdir_s, "noisy")
    lens = {}
    for ln in open(paf, "rb"):
        f = ln.split(b"\t")
        lens.setdefault(f[0], int(f[1]))
        lens.setdefault(f[5], int(f[6]))
    rnd = random.Random(11)
    fa = os.path.join(tmpdir_s, "cli_reads.fa")
    with open(fa, "w") as out:
        for nm, n in lens.items():
            out.write(">%s\n%s\n" % (nm.decode(), "".join(rnd.choice("ACGT") for _ in range(n))))
    ref_out, ref_log = R.run_cli(R.REF_BIN, ["-f", fa], paf)
    assert b"\tLN:i:" in ref_out and not ref_out.split(b"\n")[0].split(b"\t")[2].startswith(b"*")
    _same(ma.CLI_PATH, ["-f", fa], paf, ref_out, ref_log, "cli -f")
    _same(R.DROPIN_BIN, ["-f", fa], paf, ref_out, ref_log, "dropin -f")


def test_cli_reads_file_shorter_than_the_paf_says(tmpdir_s):
    """-f with a reads file that disagrees with the PAF: a read shorter than the lengths in the PAF makes the reference read outside its buffer
    (asm.c:279-285, undefined).  Here the missing bases stay 'N' and nothing outside the batch is touched: no fault, same unitig lengths.
    (-1 -2: without read selection the placement uses the full PAF lengths; with it the reference -- and this library -- assert.)"""
    import random
    import subprocess
    paf = R.pafgen(os.path.join(tmpdir_s, "cli_short.paf"), 600, 20000, 21, ["-L", "uniform"])
    lens = {}
    for ln in open(paf, "rb"):
        f = ln.split(b"\t")
        lens.setdefault(f[0], int(f[1]))
        lens.setdefault(f[5], int(f[6]))
    rnd = random.Random(3)
    fa = os.path.join(tmpdir_s, "cli_short.fa")
    with open(fa, "w") as out:
        for k, (nm, n) in enumerate(lens.items()):
            n = n - 9 if k % 7 == 0 else n
            out.write(">%s\n%s\n" % (nm.decode(), "".join(rnd.choice("ACGT") for _ in range(n))))
    r = subprocess.run([ma.CLI_PATH, "-1", "-2", "-f", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    plain = subprocess.run([ma.CLI_PATH, "-1", "-2", paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    seqs = [l.split(b"\t") for l in r.stdout.split(b"\n") if l.startswith(b"S\t")]
    want = [l.split(b"\t") for l in plain.stdout.split(b"\n") if l.startswith(b"S\t")]
    assert len(seqs) == len(want) and len(seqs) > 0
    for a, b in zip(seqs, want):
        assert a[1] == b[1] and a[3] == b[3] and len(a[2]) == int(b[3].split(b":")[2])  # same unitig, same LN, a base (or N) for every position
        assert set(a[2]) <= set(b"ACGTN")


TIE_INPUTS = {  # coordinates on a grid (pafgen -q): hundreds of equal (qid,qs) hit keys and (u,len) arc keys
    "grid16": dict(reads=3000, lines=80000, seed=5, extra=["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"]),
    "grid200": dict(reads=3000, lines=80000, seed=6, extra=["-q", "200", "-L", "uniform", "-d", "0.3", "-x", "0.03"]),
    "grid400_lognormal": dict(reads=4000, lines=120000, seed=7, extra=["-q", "400", "-d", "0.2"]),
    "grid50_fixed": dict(reads=2500, lines=70000, seed=8, extra=["-q", "50", "-L", "fixed", "-d", "0.1"]),
    "grid400_deep": dict(reads=30000, lines=1500000, seed=9, extra=["-q", "400", "-d", "0.2", "-x", "0.03"]),  # 97 % of the reads contained (squeezed-id keys matter)
    # what a real overlapper writes (pafgen -j / -b / -t, round 6): every coordinate jittered on its own, some pairs listed from both sides, the lines grouped by
    # TARGET -- equal keys by chance, a query's records scattered over the file (the sort cannot take runs), both readers' dictionaries fed in an order of their own
    "jitter_by_target": dict(reads=3000, lines=80000, seed=11, extra=["-j", "30", "-b", "0.1", "-t", "-L", "uniform", "-d", "0.2", "-x", "0.03"]),
    "jitter_both_ways": dict(reads=3000, lines=80000, seed=11, extra=["-j", "8", "-b", "0.2", "-d", "0.15"]),
}


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("mode", ["default", "forced"])
@pytest.mark.parametrize("name", list(TIE_INPUTS))
def test_tie_rich_input_is_byte_identical(name, mode, tmpdir_s, monkeypatch):
    """On inputs full of equal sort keys every dump equals the reference's BYTE FOR BYTE -- no line-order normalisation: hit
    order, arc order and everything downstream of them.  "default": nothing set -- the tie census after the arc sort finds
    the tie groups and the reference's order is reproduced automatically; "forced": MA_EXACT_TIES=1 (mahip_set_exact_ties 1)"""
    if mode == "forced":
        monkeypatch.setenv("MA_EXACT_TIES", "1")
    else:
        monkeypatch.delenv("MA_EXACT_TIES", raising=False)
    cfg = TIE_INPUTS[name]
    paf = R.pafgen(os.path.join(tmpdir_s, "tie_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
    assert R.arc_tie_groups(ref_sg) >= 5, "input is supposed to be tie-rich"
    for args in DUMPS:
        ref_out, ref_log = R.run_cli(R.REF_BIN, args, paf)
        out, log = R.run_cli(ma.CLI_PATH, args, paf)
        assert out == ref_out, "cli[%s] %s: bytes differ from the reference" % (name, " ".join(args))
        assert [x for x in R.counters(log) if not x.startswith("main: Version")] == R.counters(ref_log)
    for args in (["-p", "paf"], ["-p", "sg", "-S5"], ["-p", "ug"]):
        ref_out, _ = R.run_cli(R.REF_BIN, args, paf)
        out, _ = R.run_cli(R.DROPIN_BIN, args, paf)
        assert out == ref_out, "dropin[%s] %s: bytes differ from the reference" % (name, " ".join(args))
    # unfused resident pipeline as well
    monkeypatch.setenv("MA_NO_FUSE", "1")
    ref_out, _ = R.run_cli(R.REF_BIN, ["-p", "ug"], paf)
    out, _ = R.run_cli(ma.CLI_PATH, ["-p", "ug"], paf)
    assert out == ref_out


UNSEEN_INPUTS = [  # (reads, lines, seed, extra): arc tie groups AND push conflicts, none of them in sight of the reference's arc sort (csrc/graph.hip: k_arc_push_conflicts_seen)
    (400, 9000, 8, ["-j", "8", "-b", "0.2", "-d", "0.15"]), (400, 9000, 10, ["-j", "3", "-d", "0.2", "-x", "0.03", "-L", "uniform"]), (800, 20000, 2, ["-j", "8", "-b", "0.2", "-d", "0.15"]),
    (800, 20000, 9, ["-j", "8", "-b", "0.2", "-d", "0.15"]), (1000, 25000, 8, ["-j", "30", "-b", "0.1", "-t"])]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", range(len(UNSEEN_INPUTS)))
def test_conflicts_out_of_sight_skip_the_hit_walk(case, tmpdir_s, monkeypatch):
    """Two arcs pushed from hits with equal (qid,qs) are a push conflict: only the reference's unstable hit sort knows which comes first.  The arc sort can turn that into a
    difference only if the two arcs part in a bucket of its radix passes that is walked (longer than the insertion-sort cut-off, ksort.h:182) and holds a group of equal
    keys, or in an insertion-sorted bucket where one of the two has a twin (csrc/graph.hip: k_arc_push_conflicts_seen).  These inputs have tie groups (the arc walk runs)
    and conflicts -- all of them out of sight:
    the walk over the hit keys is skipped and every dump still equals the reference's byte for byte; with the filter off (MA_TIE_NO_FILTER=1) the walk runs and gives the
    same bytes."""
    import re
    reads, lines, seed, extra = UNSEEN_INPUTS[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "unseen_%d.paf" % case), reads, lines, seed, extra)
    monkeypatch.setenv("MA_PIPE_TIMING", "1")
    for filt in (True, False):
        if filt:
            monkeypatch.delenv("MA_TIE_NO_FILTER", raising=False)
        else:
            monkeypatch.setenv("MA_TIE_NO_FILTER", "1")
        for args in (["-p", "sg", "-S5"], ["-p", "sg"], ["-p", "ug"]):
            ref_out, _ = R.run_cli(R.REF_BIN, args, paf)
            out, log = R.run_cli(ma.CLI_PATH, args, paf)
            assert out == ref_out, "case %d %s (filter %s): bytes differ from the reference" % (case, " ".join(args), filt)
            m = re.search(r"\[T::ties\] (\d+) arc tie groups \(\d+ arcs\), (\d+) push conflicts \((\d+) of them in sight of the arc sort\) -> arc walk (\d), hit walk (\d)", log)
            assert m, log[-500:]
            groups, conf, seen, arc_walk, hit_walk = (int(x) for x in m.groups())
            assert groups > 0 and conf > 0 and arc_walk == 1, "the input is supposed to have tie groups and conflicts"
            assert (seen, hit_walk) == ((0, 0) if filt else (conf, 1)), (seen, hit_walk, conf)


SEEN_INPUTS = [  # (reads, lines, seed, extra): conflicts IN sight, in a few of the reads
    (3000, 80000, 5, ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"]), (20000, 600000, 11, ["-j", "30", "-b", "0.1", "-t"]), (4000, 120000, 7, ["-q", "400", "-d", "0.2"]),
    (30000, 900000, 12, ["-j", "8", "-b", "0.5"])]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", range(len(SEEN_INPUTS)))
def test_hit_walk_takes_its_order_for_the_reads_in_sight_only(case, tmpdir_s, monkeypatch):
    """With conflicts in sight the walk over the hit keys runs, but its order is taken inside the hit groups of the reads that HAVE such a conflict only (host/refsort.c:
    ma_refsort_packed_wanted sorts the buckets below a level only where one of those reads is; every other read keeps the stable order): same bytes as the reference, and
    the same as with the whole order taken (MA_TIE_WALK_ALL=1)."""
    import re
    reads, lines, seed, extra = SEEN_INPUTS[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "seen_%d.paf" % case), reads, lines, seed, extra)
    monkeypatch.setenv("MA_PIPE_TIMING", "1")
    for args in (["-p", "sg"], ["-p", "ug"]):
        ref_out, _ = R.run_cli(R.REF_BIN, args, paf)
        for walk_all in (False, True):
            if walk_all:
                monkeypatch.setenv("MA_TIE_WALK_ALL", "1")
            else:
                monkeypatch.delenv("MA_TIE_WALK_ALL", raising=False)
            out, log = R.run_cli(ma.CLI_PATH, args, paf)
            assert out == ref_out, "case %d %s (walk all: %s): bytes differ from the reference" % (case, " ".join(args), walk_all)
            m = re.search(r"\[T::ties\] (\d+) arc tie groups .* \((\d+) of them in sight of the arc sort\) -> arc walk 1, hit walk 1( \(its order taken for (\d+) reads\))?", log)
            assert m, log[-500:]
            assert int(m.group(2)) > 0
            if walk_all:
                assert m.group(3) is None
            else:
                assert m.group(3) and 0 < int(m.group(4)) <= int(m.group(2)) and int(m.group(4)) < reads // 4
    monkeypatch.delenv("MA_TIE_WALK_ALL", raising=False)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("form", ["MA_REFSORT_TOP_APART", "MA_REFSORT_KEYS"])
def test_tie_walk_transport_forms(form, tmpdir_s, monkeypatch):
    """how the keys travel to the host walk and back (csrc/radix.hip: reference_order): packed on the device into one word (the default, covered by every
    tie test), packed WITHOUT the digit of the top level, which goes down as a byte array (keys too wide for a word: BASELINE configs[4]; forced here), or
    as raw keys that the host packs (wider still).  Same bytes as the reference in every form, hit dump included."""
    monkeypatch.setenv(form, "1")
    cfg = TIE_INPUTS["short_reads"] if "short_reads" in TIE_INPUTS else list(TIE_INPUTS.values())[0]
    paf = R.pafgen(os.path.join(tmpdir_s, "tie_form.paf"), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    for args in ([], ["-p", "paf"], ["-p", "sg", "-S5"]):
        ref_out, _ = R.run_cli(R.REF_BIN, args, paf)
        out, _ = R.run_cli(ma.CLI_PATH, args, paf)
        assert out == ref_out, "%s=1, cli %s: bytes differ from the reference" % (form, " ".join(args))


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_tie_census_and_walks_are_reported(tmpdir_s):
    """the C ABI reports what the automatic mode found and did (mahip_tie_stats): tie groups -> arc walk (and the hit walk
    only when two arcs were pushed from hits with equal keys); a tie-free input -> nothing to do"""
    opt = ma.default_opt()
    for name, cfg, want_ties in (("grid16", TIE_INPUTS["grid16"], True), ("lognormal", INPUTS["lognormal"], False)):
        paf = R.pafgen(os.path.join(tmpdir_s, "ts_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
        ing = ma.Ingest(paf, opt)
        ctx = ma.Ctx(0)
        ctx.set_exact_ties(2)
        ctx.hits_upload(ing.hits, ing.n_seq)
        gfa = ma.run_resident(ctx, opt, ing, "ug")
        st = ctx.tie_stats()
        ref_gfa, _ = R.run_cli(R.REF_BIN, [], paf)
        assert gfa == ref_gfa
        if want_ties:
            assert st["arc_tie_groups"] >= 5 and st["arc_tie_arcs"] >= 2 * st["arc_tie_groups"] and st["arc_walk"] == 1 and st["unrepaired"] == 0
            assert st["hit_walk"] == (1 if st["push_conflicts_seen"] else 0) and st["push_conflicts_seen"] <= st["push_conflicts"]
        else:
            assert st["arc_tie_groups"] == 0 and st["arc_walk"] == 0 and st["hit_walk"] == 0
        # mode 0 keeps the stable order and says so
        ctx.set_exact_ties(0)
        ctx.hits_upload(ing.hits, ing.n_seq)
        ma.run_resident(ctx, opt, ing, "ug")
        st0 = ctx.tie_stats()
        assert st0["arc_walk"] == 0 and st0["hit_walk"] == 0
        ing.close()
        ctx.close()


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_cli_degenerate_inputs(tmpdir_s, monkeypatch):
    """empty file, junk only, one line, everything filtered out, reads that all contain each other: same (mostly empty) output,
    same counters and exit status as the reference, with the device parser and with the host reader"""
    tiny = R.pafgen(os.path.join(tmpdir_s, "deg_tiny.paf"), 50, 200, 9, [])
    first = open(tiny, "rb").readline()
    cases = {"empty": b"", "junk": b"junk\tline\nmore\n\n\n", "one": first, "one_nonl": first.rstrip(b"\n"),
             "dup": first * 5, "tiny": open(tiny, "rb").read()}
    for name, blob in cases.items():
        p = os.path.join(tmpdir_s, "deg_%s.paf" % name)
        open(p, "wb").write(blob)
        for args in (["-p", "ug"], ["-p", "sg"], ["-p", "paf"], ["-p", "bed"], ["-s", "100000"], ["-m", "1000000", "-p", "paf"]):
            ref_out, ref_log = R.run_cli(R.REF_BIN, args, p)
            for host_parse in (False, True):
                if host_parse:
                    monkeypatch.setenv("MA_HOST_PARSE", "1")
                else:
                    monkeypatch.delenv("MA_HOST_PARSE", raising=False)
                _same(ma.CLI_PATH, args, p, ref_out, ref_log, "cli[%s,%s]" % (name, "host" if host_parse else "device"))
            monkeypatch.delenv("MA_HOST_PARSE", raising=False)
        ref_out, ref_log = R.run_cli(R.REF_BIN, ["-p", "ug"], p)
        _same(R.DROPIN_BIN, ["-p", "ug"], p, ref_out, ref_log, "dropin[%s]" % name)


# ---- BASELINE-scale inputs (configs[1]: 10 M overlaps / 200 k reads): lognormal (containment-heavy), fixed-length (20 M arcs: the
# graph kernels see real work), noisy (tips, bubbles, short overlaps: the device cleaners at scale) -- against the reference binary
BIG_INPUTS = {
    "cfg2_lognormal": dict(reads=200000, lines=10000000, seed=1, extra=[]),
    "cfg2_fixed": dict(reads=200000, lines=10000000, seed=11, extra=["-L", "fixed"]),
    "cfg2_noisy": dict(reads=400000, lines=10000000, seed=12, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    # (the 50.8 M-overlap noisy input -- arc tie groups + push conflicts, probes beyond the first table tier -- is compared by digest below: DIGEST_INPUTS)
}


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", list(BIG_INPUTS))
def test_cli_matches_reference_at_baseline_scale(name, tmpdir_s):
    import hashlib
    import subprocess
    cfg = BIG_INPUTS[name]
    paf = R.pafgen(os.path.join(tmpdir_s, "big_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    for args in cfg.get("dumps", (["-p", "sg", "-S6"], ["-p", "ug"])):
        digests = []
        for binary in (R.REF_BIN, ma.CLI_PATH):
            r = subprocess.run([binary] + args + [paf], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=1200)
            assert r.returncode == 0
            digests.append((hashlib.md5(r.stdout).hexdigest(), len(r.stdout)))
        assert digests[0] == digests[1], "%s %s: bytes differ from the reference (raw md5, no normalisation)" % (name, " ".join(args))
        assert digests[0][1] > 1000
    os.remove(paf)


# ---- the largest BASELINE configurations by DIGEST (round-4 review, row J2): configs[2] stand-in (40 M overlaps), the 50 M noisy input, the graph-heavy
# 100 M-overlap input (200 M arcs), BASELINE configs[3] (100 M) and configs[4] (500 M overlaps, high-repeat: tie walk, tier-1 bubble tables, 67-bit packed keys all
# live at once).  The reference alone needs 1 - 10 minutes and up to 50 GB on them, so it ran ONCE, in the build container (tests/golden/make_big.py ->
# tests/golden/big.json: pafgen arguments, digest of the text, RAW md5 + size of the reference's GFA); here the seeded generator writes the same text again (checked)
# and the command line's GFA is digested while it streams out.  On by default; MA_TEST_BIG_SKIP=cfg5,... leaves entries out, MA_TEST_BIG_DIR names a directory
# with 35 GB of room for the text of configs[4] (default: the test's temporary directory).
DIGEST_INPUTS = ["cfg3", "noisy50", "cfg4", "graph", "cfg5", "real10", "real50"]  # real10 / real50 (round 6): jittered coordinates, pairs from both sides, lines grouped by target


@pytest.mark.parametrize("name", DIGEST_INPUTS)
def test_cli_digest_of_the_largest_configurations(name, tmpdir_s):
    import shutil
    if name in os.environ.get("MA_TEST_BIG_SKIP", "").split(","):
        pytest.skip("left out by MA_TEST_BIG_SKIP")
    if getattr(ma, "IS_EMU", False):
        pytest.skip("the CPU build of the kernels is not meant for 40 M+ overlaps")
    gold = R.big_golden()[name]
    cfg = gold["pafgen"]
    where = os.environ.get("MA_TEST_BIG_DIR", tmpdir_s)
    if shutil.disk_usage(where).free < gold["paf_bytes"] + (2 << 30):
        pytest.skip("not enough room in %s for %d bytes of PAF text" % (where, gold["paf_bytes"]))
    paf = R.pafgen(os.path.join(where, "digest_%s.paf" % name), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    try:
        assert os.path.getsize(paf) == gold["paf_bytes"] and R.head_tail_md5(paf) == gold["paf_head_tail_md5"], "%s: the generator did not reproduce the recorded text" % name
        md5, n, log = R.md5_of_stdout([ma.CLI_PATH, paf], timeout=1800)
        assert (md5, n) == (gold["gfa_md5"], gold["gfa_bytes"]), "%s: GFA differs from the reference's (raw md5 %s / %d bytes, recorded %s / %d)" % (name, md5, n, gold["gfa_md5"], gold["gfa_bytes"])
    finally:
        os.remove(paf)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_bubble_probes_that_outgrow_their_tables(tmpdir_s, monkeypatch):
    """a bubble probe that fills its table hands its source to a launch with bigger tables (csrc/clean.hip, tiers); with tier 0
    shrunk to 4 slots nearly every probe of this input takes that road, twice, and the output must not change"""
    paf = _gen(tmpdir_s, "noisy")
    monkeypatch.setenv("MA_BUBBLE_CAP0", "4")
    monkeypatch.setenv("MA_PIPE_TIMING", "2")
    # the tiers above the first probe with a wave per source, the table in LDS while it fits (default) or in HBM (forced here by a tiny limit);
    # MA_BUBBLE_THREAD_TIERS: a thread per source in every tier, the form of round 2
    for extra in ({}, {"MA_BUBBLE_LDS_CAP": "16"}, {"MA_BUBBLE_THREAD_TIERS": "1"}):
        for k in ("MA_BUBBLE_LDS_CAP", "MA_BUBBLE_THREAD_TIERS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        for args in ([], ["-p", "sg"]):
            ref_out, _ = R.run_cli(R.REF_BIN, args, paf)
            out, log = R.run_cli(ma.CLI_PATH, args, paf)
            assert out == ref_out, "%r %r" % (extra, args)
            assert "(tier 2)" in log, "the input was supposed to reach the third tier of probe tables"
