"""Device-side PAF ingest (csrc/paf.hip: lines, columns, strtol numbers, filter, name dictionary with first-appearance
ids, mirrored records) against the host reader -- which tests/test_host_vs_ref.py pins to the reference's paf.c /
sdict.c / hit.c:70-101 on the CPU -- and against the reference library itself: same records in the same order, same
names, same ids, same first-seen lengths, same counters."""
import ctypes as C
import gzip
import os
import random

import numpy as np
import pytest

import miniasm_amd as ma
import refapi as R

pytestmark = pytest.mark.gpu


def _same_as_host(ctx, path, opt=None, bi_dir=True):
    host = ma.Ingest(path, opt, bi_dir)
    dev = ma.GpuIngest(ctx, path, opt, bi_dir)
    assert dev.n == host.n and dev.n_seq == host.n_seq, (dev.n, host.n, dev.n_seq, host.n_seq)
    assert dev.names() == host.names()
    assert list(dev.lens()) == list(host.lens())
    assert dev.hits.tobytes() == host.hits.tobytes(), "records differ (or their order)"
    host.close(); dev.close()
    return dev.n


def _adversarial(path, seed=5):
    """every reader quirk the reference has: CRLF, 10-column lines (stale bl), short/junk/empty lines, signs, blanks,
    junk after digits, numbers that overflow 32 and 64 bits, 13+ columns, a NUL inside a name, names that are prefixes of
    each other, self hits, no newline at the end"""
    rnd = random.Random(seed)
    names = ["r%d" % i for i in range(300)] + ["r1", "r10", "r100", "r1000", "x", "xx", "x" * 300, "read with space", "r\x001"]
    # names around 8 / 16 bytes (the comparison works on 8-byte words), names that share a long prefix and differ behind it, a name that IS another one's prefix
    names += ["m54119_180101_0001", "m54119_180101_0002", "m54119_180101_00", "m54119_180101_000", "abcdefghijklmnop", "abcdefghijklmnopq", "abcdefghijklmnopX",
              "abcdefghijklmno", "abcdefgh", "abcdefghi", "abcdefg", "abcdefghijklmnop" * 4, "abcdefghijklmnop" * 4 + "z"]
    lines = []
    for k in range(40000):
        q, t = rnd.choice(names), rnd.choice(names)
        ql, tl = rnd.randint(3000, 20000), rnd.randint(3000, 20000)
        qs = rnd.randint(0, ql // 2); qe = rnd.randint(qs, ql)
        ts = rnd.randint(0, tl // 2); te = rnd.randint(ts, tl)
        ml = rnd.randint(0, 3000); bl = rnd.randint(ml, ml + 5000)
        f = [q, str(ql), str(qs), str(qe), rnd.choice("+-"), t, str(tl), str(ts), str(te), str(ml), str(bl), "255"]
        x = rnd.random()
        if x < 0.03: f = f[:10]                                   # stale bl
        elif x < 0.05: f = f[:rnd.randint(1, 9)]                   # skipped
        elif x < 0.06: f = []                                      # empty line
        elif x < 0.08: f[2] = " " + f[2]; f[3] = "+" + f[3]        # strtol blanks and sign
        elif x < 0.09: f[7] = "-" + f[7]                           # negative -> wraps
        elif x < 0.10: f[9] = f[9] + "abc"                         # junk after digits
        elif x < 0.105: f[1] = "99999999999"                       # > 32 bits: truncated
        elif x < 0.11: f[6] = "99999999999999999999999"            # > 64 bits: saturates
        elif x < 0.115: f[10] = "-99999999999999999999999"
        elif x < 0.12: f[4] = ""                                   # empty strand column
        elif x < 0.13: f += ["tp:A:P", "cm:i:5"]
        elif x < 0.135: f[8] = "x12"                               # no digits -> 0
        ln = "\t".join(f)
        if rnd.random() < 0.05: ln += "\r"
        lines.append(ln)
    with open(path, "wb") as g:
        g.write("\n".join(lines).encode("latin-1"))                # no trailing newline
    return path


def test_ingest_clean_inputs(tmpdir_s):
    ctx = ma.Ctx(0)
    for reads, lines, seed, extra in ((3000, 80000, 41, []), (20000, 1000000, 3, ["-L", "uniform", "-d", "0.2"]), (50, 200, 9, []),
                                      (3000, 80000, 42, ["-N", "m64011_190830_220126/"]), (20000, 1000000, 4, ["-N", "m64011_190830_220126/", "-j", "20", "-b", "0.1", "-t"])):  # names of the length real files carry: the text-comparing dictionary, the hashed keys
        paf = R.pafgen(os.path.join(tmpdir_s, "gi_%d.paf" % seed), reads, lines, seed, extra)
        assert _same_as_host(ctx, paf) > 0
        _same_as_host(ctx, paf, bi_dir=False)
    ctx.close()


def test_ingest_adversarial_text(tmpdir_s):
    ctx = ma.Ctx(0)
    paf = _adversarial(os.path.join(tmpdir_s, "gi_adv.paf"))
    n = _same_as_host(ctx, paf)
    assert n > 1000
    opt = ma.default_opt(); opt.min_span = 500; opt.min_match = 10
    _same_as_host(ctx, paf, opt)
    # with a trailing newline, with only junk, empty, one line
    data = open(paf, "rb").read()
    for k, blob in enumerate((data + b"\n", b"junk\tline\nmore junk\n", b"", b"\n\n\n", data.split(b"\n")[0], data.split(b"\n")[0] + b"\n")):
        p = os.path.join(tmpdir_s, "gi_edge%d.paf" % k)
        open(p, "wb").write(blob)
        _same_as_host(ctx, p, opt)
    ctx.close()


def test_name_table_growth(tmpdir_s, monkeypatch):
    """the name table starts small (sized for 16 lines per name) and the insert pass is repeated with a larger one when the
    load factor comes out above 1/2 or a probe sequence runs out: force both from tiny start sizes"""
    ctx = ma.Ctx(0)
    adv = _adversarial(os.path.join(tmpdir_s, "gi_grow_adv.paf"))
    paf = R.pafgen(os.path.join(tmpdir_s, "gi_grow.paf"), 3000, 80000, 41, [])
    for log2 in ("4", "8", "11"):
        monkeypatch.setenv("MA_DICT_CAP_LOG2", log2)
        assert _same_as_host(ctx, adv) > 1000
        assert _same_as_host(ctx, paf) > 0
    monkeypatch.delenv("MA_DICT_CAP_LOG2")
    ctx.close()


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_ingest_matches_reference_library(tmpdir_s):
    """straight against the reference's ma_hit_read (sorted there, so compare canonicalised records) and its dictionary"""
    ctx = ma.Ctx(0)
    paf = _adversarial(os.path.join(tmpdir_s, "gi_adv2.paf"), seed=11)
    opt = ma.default_opt(); opt.min_span = 500; opt.min_match = 10
    LR = R.ref()
    d = LR.sd_init()
    n = C.c_size_t(0)
    q = LR.ma_hit_read(paf.encode(), opt.min_span, opt.min_match, d, C.byref(n), 1, None)
    ref_hits = R.np_from(q, n.value, ma.HIT_DT)
    ref_hits["bldel"] &= 0x7FFFFFFF
    dev = ma.GpuIngest(ctx, paf, opt)
    assert dev.n == n.value
    assert dev.names() == [d.contents.seq[i].name.decode("latin-1") for i in range(d.contents.n_seq)] or \
        [x.encode() for x in dev.names()] == [d.contents.seq[i].name for i in range(d.contents.n_seq)]
    assert list(dev.lens()) == [d.contents.seq[i].len for i in range(d.contents.n_seq)]
    assert R.canon(dev.hits).tobytes() == R.canon(ref_hits).tobytes()
    dev.close(); ctx.close()


def test_ingest_gz_and_cli_paths_agree(tmpdir_s, monkeypatch):
    """gzip input goes through the inflate-then-upload branch; the CLI output with the device parser equals the one with
    the host reader for every dump"""
    paf = R.pafgen(os.path.join(tmpdir_s, "gi_cli.paf"), 4000, 90000, 63, ["-L", "uniform", "-d", "0.35", "-x", "0.03"])
    gz = paf + ".gz"
    with open(paf, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    ctx = ma.Ctx(0)
    a = ma.GpuIngest(ctx, paf); b = ma.GpuIngest(ctx, gz)
    assert a.n == b.n and a.names() == b.names()
    a.close(); b.close(); ctx.close()
    for args in (["-p", "bed"], ["-p", "paf"], ["-p", "sg", "-S5"], ["-p", "ug"], ["-b"], ["-1", "-2", "-p", "sg"]):
        monkeypatch.setenv("MA_HOST_PARSE", "1")
        host_out, host_log = R.run_cli(ma.CLI_PATH, args, paf)
        monkeypatch.delenv("MA_HOST_PARSE")
        dev_out, dev_log = R.run_cli(ma.CLI_PATH, args, paf)
        assert dev_out == host_out, args
        assert R.counters(dev_log) == R.counters(host_log), args
    r_out, _ = R.run_cli(ma.CLI_PATH, ["-p", "ug"], gz)
    assert r_out == R.run_cli(ma.CLI_PATH, ["-p", "ug"], paf)[0]


def test_ingest_fuzz_against_host_reader(tmpdir_s):
    """random ASCII soup with the separators, signs, blanks, CR and NUL over-represented: whatever the host reader (pinned to
    the reference on the CPU) makes of it, the device parser must make the same of it"""
    rnd = random.Random(2024)
    alphabet = "0123456789" * 6 + "\t" * 14 + "\n" * 3 + "+- \r\x00\x0b\x0cabcxyzACGT:_.|" + "\t\t"
    ctx = ma.Ctx(0)
    opt = ma.default_opt(); opt.min_span = 0; opt.min_match = 0
    total = 0
    for k in range(12):
        n = rnd.choice((0, 1, 7, 300, 5000, 60000, 400000))
        txt = "".join(rnd.choice(alphabet) for _ in range(n))
        if k % 3 == 0:  # sprinkle well-formed lines so that the dictionary and the mirrored records are exercised too
            good = ["r%d\t9000\t%d\t%d\t%s\tr%d\t8000\t%d\t%d\t%d\t%d\t255" % (rnd.randint(0, 40), a, a + rnd.randint(0, 5000), rnd.choice("+-"), rnd.randint(0, 40), b,
                                                                         b + rnd.randint(0, 5000), rnd.randint(0, 900), rnd.randint(0, 4000))
                    for a, b in ((rnd.randint(0, 3000), rnd.randint(0, 3000)) for _ in range(400))]
            parts = txt.split("\n")
            for g in good:
                parts.insert(rnd.randint(0, len(parts)), g)
            txt = "\n".join(parts)
        p = os.path.join(tmpdir_s, "gi_fuzz%d.paf" % k)
        open(p, "wb").write(txt.encode("ascii"))
        total += _same_as_host(ctx, p, opt)
        total += _same_as_host(ctx, p, opt, bi_dir=False)
    assert total > 1000
    ctx.close()


def _good_line(rnd, names, tags=""):
    a, b = rnd.randint(0, 3000), rnd.randint(0, 3000)
    return "%s\t9000\t%d\t%d\t%s\t%s\t8000\t%d\t%d\t%d\t%d\t255%s" % (rnd.choice(names), a, a + rnd.randint(2100, 5000), rnd.choice("+-"), rnd.choice(names), b,
                                                                     b + rnd.randint(2100, 5000), rnd.randint(150, 900), rnd.randint(1000, 4000), tags)


def test_tile_parser_shapes(tmpdir_s, monkeypatch):
    """the tile parser (csrc/paf.hip: k_paf_parse_tile) cuts the text into tiles of K KiB and gives every line that ENDS in a tile a lane of its block: texts whose size
    sits on and around tile and granule borders, with and without a final newline; lines longer than the 960 bytes a block keeps in front of its tile, longer than a tile,
    longer than many tiles; thousands of empty lines in one tile (more lines than lanes: batches); long lines on average (the 32 KiB form of the kernel); names of 8, 9,
    64 and 65 bytes (key = the bytes / a hash of the words / the byte-wise routine); every tile size the host may choose, forced"""
    rnd = random.Random(77)
    short = ["r%d" % i for i in range(60)] + ["abcdefgh", "abcdefg", "x"]
    mixed = short + ["abcdefghi", "m54119_180101_0001/12345/0_9000", "n" * 64, "n" * 63 + "x", "n" * 65, "n" * 200]
    ctx = ma.Ctx(0)
    opt = ma.default_opt()
    cases = []
    base_lines = [_good_line(rnd, short) for _ in range(1200)]
    body = "\n".join(base_lines)
    for cut in (1024, 1023, 1025, 2048, 15 * 1024, 15 * 1024 + 1, 16384, 16383, 31 * 1024, 32768, 40000):
        cases.append(("cut%d" % cut, body[:cut]))
        cases.append(("cut%dnl" % cut, body[:cut - 1] + "\n"))
    # long lines: tags of 1 KB, 3 KB, 20 KB, 70 KB between ordinary lines; a long line first, a long line last
    for k, tl in enumerate((900, 1000, 3000, 20000, 70000)):
        ls = [_good_line(rnd, mixed) for _ in range(300)]
        for _ in range(6):
            ls.insert(rnd.randint(0, len(ls)), _good_line(rnd, mixed, "\tcg:Z:" + "M" * tl))
        ls.insert(0, _good_line(rnd, mixed, "\tcg:Z:" + "I" * tl))
        ls.append(_good_line(rnd, mixed, "\tcg:Z:" + "D" * tl))
        cases.append(("long%d" % k, "\n".join(ls) + ("\n" if k & 1 else "")))
    # more lines than lanes in a tile: runs of empty lines, one-byte lines
    cases.append(("empties", "\n" * 5000 + body[:3000] + "\n" * 3000 + "\n".join(base_lines[:40]) + "\n" + "x\n" * 2000))
    # long on average: 30 bytes of tags on every line -> K > 15
    cases.append(("tagged", "\n".join(_good_line(rnd, mixed, "\ttp:A:S\tcm:i:%d\ts1:i:%d\tdv:f:0.0123\trl:i:55" % (rnd.randint(0, 999), rnd.randint(0, 9999))) for _ in range(3000)) + "\n"))
    cases.append(("mixednames", "\n".join(_good_line(rnd, mixed) for _ in range(4000))))
    for name, txt in cases:
        p = os.path.join(tmpdir_s, "gi_tile_%s.paf" % name)
        open(p, "wb").write(txt.encode("latin-1"))
        _same_as_host(ctx, p, opt)
    for k in ("1", "2", "7", "15", "16", "31"):
        monkeypatch.setenv("MA_PAF_TILE_K", k)
        for name in ("long3", "empties", "tagged", "cut16384", "cut32768nl"):
            _same_as_host(ctx, os.path.join(tmpdir_s, "gi_tile_%s.paf" % name), opt)
    monkeypatch.delenv("MA_PAF_TILE_K")
    # round 5's kernels stay behind a switch: same result
    monkeypatch.setenv("MA_PAF_OLD", "1")
    for name in ("long3", "mixednames", "empties"):
        _same_as_host(ctx, os.path.join(tmpdir_s, "gi_tile_%s.paf" % name), opt)
    monkeypatch.delenv("MA_PAF_OLD")
    # short names only, but the text-comparing dictionary forced: same ids
    monkeypatch.setenv("MA_DICT_EXACT_TEXT", "1")
    _same_as_host(ctx, os.path.join(tmpdir_s, "gi_tile_cut40000.paf"), opt)
    monkeypatch.delenv("MA_DICT_EXACT_TEXT")
    ctx.close()


def test_cli_falls_back_to_host_reader_when_text_stage_does_not_fit(tmpdir_s, monkeypatch):
    """a text the device stage refuses (here: an artificial cap) is parsed by the host reader instead: same output, a warning"""
    import subprocess
    paf = R.pafgen(os.path.join(tmpdir_s, "gi_fb.paf"), 3000, 80000, 41, [])
    base, _ = R.run_cli(ma.CLI_PATH, ["-p", "ug"], paf)
    monkeypatch.setenv("MA_PAF_MAX_BYTES", "1000")
    out, log = R.run_cli(ma.CLI_PATH, ["-p", "ug"], paf)
    assert out == base and "using the host reader" in log
    r = subprocess.run([R.DROPIN_BIN, "-p", "ug", paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE) if R.have_ref() else None
    if r is not None:
        assert r.returncode == 0 and r.stdout == base and b"using the host reader" in r.stderr


def _dev_buffer(nbytes):
    """device memory for the C ABI: a torch tensor on the GPU; on the CPU build of the kernels (tests/emu) device pointers are host pointers"""
    if getattr(ma, "IS_EMU", False):
        a = np.zeros(max(nbytes, 1), dtype=np.uint8)
        return a, a.ctypes.data
    import torch
    t = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
    return t, t.data_ptr()


@pytest.mark.parametrize("case", ["lognormal", "noisy"])
def test_bench_shaped_steps_from_resident_records(case, tmpdir_s):
    """the sequence bench.py times, through the same C entry points: file -> HBM, device parse, the unsorted records carved out by read range
    (mahip_hits_raw_extract) into caller-owned device memory, then per step adopt + hint + head + tail; every step's GFA equals the CLI's and the
    reference's; the ranges of a 3-way split are disjoint, in input order, and add up to the whole"""
    L = ma.lib()
    vp = C.c_void_p
    extra = {"lognormal": [], "noisy": ["-L", "uniform", "-d", "0.35", "-x", "0.03"]}[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "bs_%s.paf" % case), 3000, 90000, 61, extra)
    opt = ma.default_opt()
    L.ma_paf_load_file.argtypes = [vp, C.c_char_p]
    L.ma_hit_ingest_loaded.argtypes = [vp, C.c_int, C.c_int, C.POINTER(ma.Sdict), C.POINTER(C.c_size_t), C.c_int, C.c_int]
    L.mahip_hits_raw_extract.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_size_t)]
    L.mahip_paf_max_qs.restype = C.c_uint32
    L.mahip_paf_max_qs.argtypes = [vp]
    L.ma_pipeline_head.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
    ctx = ma.Ctx(0)
    assert L.ma_paf_load_file(ctx.h, paf.encode()) == 0
    d = L.sd_init()
    nh = C.c_size_t(0)
    assert L.ma_hit_ingest_loaded(ctx.h, opt.min_span, opt.min_match, d, C.byref(nh), 1, 1) == 0
    n_seq, max_qs = d.contents.n_seq, L.mahip_paf_max_qs(ctx.h)
    host = ma.Ingest(paf, opt)  # the host reader's records: the same records in the same order
    assert nh.value == host.n and n_seq == host.n_seq

    def extract(q0, q1):
        n = C.c_size_t(0)
        ma._chk(L.mahip_hits_raw_extract(ctx.h, q0, q1, None, C.byref(n)), "raw_extract")
        keep, ptr = _dev_buffer(n.value * 32)
        ma._chk(L.mahip_hits_raw_extract(ctx.h, q0, q1, vp(ptr), C.byref(n)), "raw_extract")
        L.mahip_sync(ctx.h)
        out = np.zeros(n.value, dtype=ma.HIT_DT)
        if n.value:
            ma._chk(L.mahip_memcpy_d2h(ctx.h, out.ctypes.data, vp(ptr), n.value * 32), "d2h")
        return keep, ptr, out

    keep_all, ptr_all, all_recs = extract(0, 0xffffffff)
    assert all_recs.tobytes() == host.hits.tobytes()
    per = (n_seq + 2) // 3
    parts = [extract(min(r * per, n_seq), min((r + 1) * per, n_seq))[2] for r in range(3)]
    assert sum(len(p) for p in parts) == len(all_recs)
    qid = (host.hits["qns"] >> np.uint64(32)).astype(np.int64)
    for r, p in enumerate(parts):
        assert p.tobytes() == host.hits[(qid >= r * per) & (qid < (r + 1) * per)].tobytes()
    # ... and mahip_hits_raw_extract_pos also says where each record of a range stood in the input (what an own-records rank hands to mahip_hits_set_positions)
    L.mahip_hits_raw_extract_pos.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(C.c_size_t)]
    for r in range(3):
        sel = (qid >= r * per) & (qid < (r + 1) * per)
        n = C.c_size_t(0)
        keep_r, ptr_r = _dev_buffer(max(int(sel.sum()), 1) * 32)
        keep_p, ptr_p = _dev_buffer(max(int(sel.sum()), 1) * 4)
        ma._chk(L.mahip_hits_raw_extract_pos(ctx.h, min(r * per, n_seq), min((r + 1) * per, n_seq), vp(ptr_r), vp(ptr_p), C.byref(n)), "raw_extract_pos")
        L.mahip_sync(ctx.h)
        pos = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            ma._chk(L.mahip_memcpy_d2h(ctx.h, pos.ctypes.data, vp(ptr_p), n.value * 4), "d2h")
        assert n.value == int(sel.sum()) and (pos == np.nonzero(sel)[0]).all()
    want, _ = R.run_cli(ma.CLI_PATH, [], paf)
    for step in range(3):
        ma._chk(L.mahip_hits_adopt(ctx.h, vp(ptr_all), len(all_recs), n_seq), "adopt")
        L.mahip_set_hints(ctx.h, max_qs)
        st = (C.c_uint32 * 4)(0, 0, 0, 0)
        assert L.ma_pipeline_head(ctx.h, C.byref(opt), d, b"ug", 100, 0, C.byref(st)) == 0
        buf, ln = vp(0), C.c_size_t(0)
        assert L.ma_pipeline_tail_mem(ctx.h, C.byref(opt), d, b"ug", 100, C.byref(st), C.byref(buf), C.byref(ln)) == 0
        got = C.string_at(buf, ln.value)
        L.free_buf(buf)
        assert got == want, "step %d differs from the CLI" % step
    if R.have_ref():
        ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
        if R.arc_tie_groups(ref_sg) == 0:
            assert want == R.run_cli(R.REF_BIN, [], paf)[0]
    L.sd_destroy(d)
    host.close()
    ctx.close()
    del keep_all
