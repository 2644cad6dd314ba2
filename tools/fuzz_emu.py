#!/usr/bin/env python3
"""Randomised differential test WITHOUT a GPU: random pafgen inputs (incl. tie-rich grids, noise, tiny and deep reads) and random
command-line options, the unmodified reference binary (oracle/_ref/miniasm_ref) against the CPU build of the kernel sources
(tests/emu/_build/miniasm), every dump format, bytes compared.  Test tooling: `python tools/fuzz_emu.py --cases 200 --seed 1`.
With --ranks the same case is also run as MA_GPUS=N over the shared-memory double."""
import argparse
import hashlib
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")
EMU = os.path.join(ROOT, "tests", "emu", "_build", "miniasm")
PAFGEN = os.path.join(ROOT, "miniasm_amd", "bin", "pafgen")
TIES = False
REAL = False


def run(binary, args, paf, env=None, stdin=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([binary] + args + ([paf] if paf else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, stdin=stdin, timeout=300)
    return r.returncode, r.stdout, r.stderr


def rand_case(rng):
    reads = rng.choice([3, 20, 60, 200, 500, 1500, 4000])
    depth = rng.choice([2, 5, 10, 25, 60, 150])
    lines = max(4, min(reads * depth // 2, 120000))
    gen = ["-r", str(reads), "-n", str(lines), "-s", str(rng.randrange(1 << 30))]
    gen += rng.choice([[], ["-L", "fixed"], ["-L", "uniform"], ["-L", "lognormal"]])
    if rng.random() < 0.5:
        gen += ["-d", "%.2f" % rng.choice([0.05, 0.2, 0.35, 0.6])]
    if rng.random() < 0.5:
        gen += ["-x", "%.3f" % rng.choice([0.005, 0.03, 0.1])]
    if rng.random() < 0.3:
        gen += ["-i", "%.2f" % rng.choice([0.1, 0.3])]
    if rng.random() < 0.3:
        gen += ["-g"]
    if REAL:  # what a real overlapper writes: jittered coordinates (equal keys by chance), pairs from both sides, lines grouped by target
        gen += ["-j", str(rng.choice([1, 3, 8, 30, 100]))]
        if rng.random() < 0.6:
            gen += ["-b", "%.2f" % rng.choice([0.05, 0.2, 0.5, 1.0])]
        if rng.random() < 0.5:
            gen += ["-t"]
    elif rng.random() < (1.0 if TIES else 0.4):
        gen += ["-q", str(rng.choice([4, 16, 64, 256]))]  # coordinates on a grid: equal sort keys everywhere
    if rng.random() < 0.3:
        gen += ["-m", str(rng.choice([1500, 3000, 20000]))]
    opts = []
    if rng.random() < 0.3:
        opts += ["-c", str(rng.choice([1, 2, 5]))]
    if rng.random() < 0.3:
        opts += ["-m", str(rng.choice([0, 50, 500]))]
    if rng.random() < 0.3:
        opts += ["-s", str(rng.choice([200, 1000, 3000]))]
    if rng.random() < 0.2:
        opts += ["-i", "%.2f" % rng.choice([0.0, 0.1, 0.3])]
    if rng.random() < 0.3:
        opts += ["-o", str(rng.choice([500, 2000]))]
    if rng.random() < 0.3:
        opts += ["-h", str(rng.choice([100, 1000, 5000]))]
    if rng.random() < 0.2:
        opts += ["-I", "%.2f" % rng.choice([0.5, 0.9])]
    if rng.random() < 0.2:
        opts += ["-g", str(rng.choice([0, 10, 5000]))]
    if rng.random() < 0.2:
        opts += ["-d", str(rng.choice([1000, 200000]))]
    if rng.random() < 0.2:
        opts += ["-e", str(rng.choice([1, 2, 10]))]
    if rng.random() < 0.2:
        opts += ["-n", str(rng.choice([1, 2, 5]))]
    if rng.random() < 0.2:
        opts += ["-r", rng.choice(["0.9,0.5", "0.6,0.3", "0.8"])]
    if rng.random() < 0.15:
        opts += ["-F", "%.1f" % rng.choice([0.5, 0.9])]
    for f in ("-1", "-2", "-b", "-B", "-R"):
        if rng.random() < 0.15:
            opts.append(f)
    dump = rng.choice([[], [], ["-p", "ug"], ["-p", "sg"], ["-p", "bed"], ["-p", "paf"], ["-p", "sg", "-S%d" % rng.randrange(1, 7)], ["-p", "paf", "-S%d" % rng.randrange(1, 5)],
                       ["-p", "bed", "-S%d" % rng.randrange(1, 4)]])
    return gen, opts + dump


def mutate_text(rng, data):
    """line- and byte-level damage to a PAF text: the reference's reader (paf.c / kseq.h / strtol semantics) is the ground truth"""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    n_mut = rng.choice([1, 3, 10, 50])
    for _ in range(n_mut):
        if not lines:
            break
        i = rng.randrange(len(lines))
        f = lines[i].split(b"\t")
        m = rng.randrange(16)
        if m == 0 and len(f) > 3:
            del f[rng.randrange(len(f))]                                  # a column less (11 -> 10: stale bl; < 10: skipped)
        elif m == 1:
            f = f[:rng.randrange(1, len(f) + 1)]                          # truncated line
        elif m == 2 and len(f) > 4:
            k = rng.choice([1, 2, 3, 6, 7, 8, 9, 10]) % len(f)
            f[k] = rng.choice([b"-5", b"+7", b" 12", b"99999999999", b"9223372036854775808", b"-9223372036854775809", b"12x", b"x", b"", b"007", b"4294967296", b"2147483648"])
        elif m == 3:
            f[-1] = f[-1] + b"\r"                                         # CRLF
        elif m == 4:
            lines.insert(i, rng.choice([b"", b"\r", b"\t\t\t", b"#comment", b"a\tb"]))
            continue
        elif m == 5:
            lines.insert(i, lines[rng.randrange(len(lines))])             # duplicated line elsewhere
            continue
        elif m == 6 and len(f) > 5:
            f[0] = rng.choice([b"", b"r" + b"x" * 300, f[5], b"a b", b"\xff\xfe", b"r1\x00tail"])   # odd query names (NUL ends a name for the reference)
        elif m == 7 and len(f) > 5:
            f[5] = rng.choice([b"", f[0], b"T" * 70, b"r0"])
        elif m == 8 and len(f) > 4:
            f[4] = rng.choice([b"-", b"+", b"", b"-x", b"*"])
        elif m == 9:
            f = f + [b"tp:A:P", b"cm:i:%d" % rng.randrange(1000)] * rng.randrange(1, 30)   # long tails of optional tags
        elif m == 10:
            f = [x.replace(b"r", b" r") if rng.random() < 0.3 else x for x in f]
        elif m == 11 and len(f) > 10:
            f[10] = rng.choice([b"0", b"1", b"2147483647", b"2147483648", b"4294967295"])
        elif m == 12:
            f = f + [b"z" * rng.choice([100, 5000, 60000])]              # a very long line (beyond the LDS stage of its block)
        elif m == 13 and len(f) > 9:
            f[9] = rng.choice([b"0", b"99", b"100", b"-1"])
        elif m == 14:
            del lines[i]
            continue
        elif m == 15 and len(f) > 3:
            f[2], f[3] = f[3], f[2]                                       # start > end
        lines[i] = b"\t".join(f)
    out = b"\n".join(lines)
    if rng.random() < 0.8:
        out += b"\n"                                                      # otherwise: unterminated last line
    return out


def write_reads(rng, paf, path):
    """a reads file for -f (asm.c:216-290 ma_ug_seq): FASTA or FASTQ, mixed case, N's, wrapped lines, some reads missing, some longer or shorter
    than the PAF says, reads the PAF never mentions"""
    lens = {}
    for ln in open(paf, "rb"):
        f = ln.rstrip(b"\n").split(b"\t")
        if len(f) > 6:
            try:
                lens.setdefault(f[0], int(f[1]))
                lens.setdefault(f[5], int(f[6]))
            except ValueError:
                pass
    fq = rng.random() < 0.4
    wrap = rng.choice([0, 0, 60, 1000])
    names = list(lens)
    rng.shuffle(names)
    with open(path, "wb") as out:
        for nm in names + [b"not_in_paf"]:
            if rng.random() < 0.03:
                continue                                    # a read without a sequence
            n = lens.get(nm, 500)
            if rng.random() < 0.05:
                n = n + rng.choice([3, 50, 1000])            # longer than the PAF says (shorter ones make the reference read outside its buffer: undefined)
            alphabet = "ACGT" if rng.random() < 0.7 else "ACGTNacgtn"
            seq = "".join(rng.choice(alphabet) for _ in range(n)).encode()
            hdr = nm + (b" extra comment" if rng.random() < 0.3 else b"")
            if fq:
                out.write(b"@" + hdr + b"\n" + seq + b"\n+\n" + b"I" * n + b"\n")
            elif wrap:
                out.write(b">" + hdr + b"\n" + b"\n".join(seq[k:k + wrap] for k in range(0, n, wrap)) + b"\n")
            else:
                out.write(b">" + hdr + b"\n" + seq + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", action="store_true", help="unitig sequences: every case also gets a random reads file and runs with -f (ma_ug_seq; the device byte gather)")
    ap.add_argument("--text", action="store_true", help="damage the PAF text (reader / dictionary semantics) instead of varying the options")
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ranks", type=int, default=0, help="also run every case as MA_GPUS=N (shared-memory double)")
    ap.add_argument("--ties", action="store_true", help="every input on a coordinate grid (equal sort keys everywhere: tie census, push conflicts, both walks)")
    ap.add_argument("--real", action="store_true", help="realistic noise: pafgen -j (jitter) / -b (both directions) / -t (grouped by target): equal keys by chance, conflicts mostly out of the arc sort's sight")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for the runs of the CPU build (e.g. MA_HOST_PARSE=1, MA_NO_FUSE=1, MA_EXACT_TIES=1, MA_THREADS=3)")
    ap.add_argument("--keep", default=None, help="directory for failing inputs")
    ap.add_argument("--emu", default=None, help="the CPU build's miniasm (default tests/emu/_build/miniasm; point it at a copy to keep fuzzing across rebuilds)")
    a = ap.parse_args()
    global EMU, TIES, REAL
    TIES = a.ties
    REAL = a.real
    if a.emu:
        EMU = a.emu
    for p in (REF, EMU, PAFGEN):
        if not os.path.exists(p):
            sys.exit("missing " + p)
    rng = random.Random(a.seed)
    tmp = tempfile.mkdtemp(prefix="ma_fuzz_")
    paf = os.path.join(tmp, "f.paf")
    bad = skipped = 0
    paths = {}
    for k in range(a.cases):
        gen, args = rand_case(rng)
        r = subprocess.run([PAFGEN] + gen + ["-o", paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0 or not os.path.exists(paf):
            print("case %d: pafgen %s failed, skipped" % (k, " ".join(gen)))
            continue
        if a.text:
            data = mutate_text(rng, open(paf, "rb").read())
            open(paf, "wb").write(data)
            args = rng.choice([["-p", "paf", "-S2"], ["-p", "paf"], ["-p", "bed"], ["-p", "sg"], [], ["-R", "-p", "paf"], ["-B", "-p", "paf", "-S2"], ["-s", "0", "-m", "0", "-p", "paf", "-S2"]])
        if a.seq:
            reads = os.path.join(tmp, "reads.fx")
            write_reads(rng, paf, reads)
            args = [x for x in args if x not in ("-p", "sg", "paf", "bed", "ug") and not x.startswith("-S")] + ["-f", reads]
        rc0, out0, err0 = run(REF, args, paf)
        if rc0 < 0:  # the reference itself dies on this combination (e.g. -p bed -S1 dereferences the intervals before they exist): nothing to compare
            skipped += 1
            continue
        runs = [("emu", {})]
        if a.ranks > 1:  # requests the sharded head does not serve fall back to one GPU: the bytes must be the same either way
            runs.append(("emu x%d" % a.ranks, {"MA_GPUS": str(a.ranks), "MA_COMM": "shm"}))
        for name, env in runs:
            env = dict(env, MA_PIPE_TIMING="1")  # the [T::ties] line: which tie path the run took
            env.update(kv.split("=", 1) for kv in a.env)
            rc1, out1, err1 = run(EMU, args, paf, env)
            for ln in err1.decode(errors="replace").splitlines():
                if ln.startswith("[T::ties]") and "arc tie groups" in ln:
                    import re
                    mm = re.search(r"(\d+) arc tie groups .*?, (\d+) push conflicts \((\d+) of them in sight", ln)
                    if not mm:
                        continue
                    groups, conf, seen = int(mm.group(1)), int(mm.group(2)), int(mm.group(3))
                    key = "no arc ties" if groups == 0 else "arc walk" if conf == 0 else "arc walk, conflicts out of sight" if seen == 0 else "arc walk + hit walk"
                    paths[key] = paths.get(key, 0) + 1
            ok = rc0 == rc1 and out0 == out1 and b"runtime error" not in err1 and b"AddressSanitizer" not in err1  # the last clause: a sanitizer build of tests/emu (make B=_build_san CXX="g++ -fsanitize=address,undefined" CC="gcc -fsanitize=address,undefined")
            if not ok:
                bad += 1
                print("case %d MISMATCH [%s]: pafgen %s | miniasm %s | rc %d vs %d, %d vs %d bytes, md5 %s vs %s" % (
                    k, name, " ".join(gen), " ".join(args), rc0, rc1, len(out0), len(out1), hashlib.md5(out0).hexdigest()[:8], hashlib.md5(out1).hexdigest()[:8]))
                print("   emu stderr tail:", err1[-300:].decode(errors="replace").replace("\n", " | "))
                if a.keep:
                    os.makedirs(a.keep, exist_ok=True)
                    os.replace(paf, os.path.join(a.keep, "case%d.paf" % k))
        if (k + 1) % 20 == 0:
            print("%d cases, %d mismatches" % (k + 1, bad), flush=True)
    print("tie paths taken:", paths)
    print("done: %d cases, %d mismatches, %d skipped (reference crashed)" % (a.cases, bad, skipped))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
