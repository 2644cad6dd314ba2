#!/usr/bin/env python3
"""Reference digests of the LARGE BASELINE configurations -> tests/golden/big.json.

    python tests/golden/make_big.py [name ...]        (default: every entry of CONFIGS; entries already recorded are kept)

Run in the build container (or on any box that holds oracle/_ref/miniasm_ref): for every configuration the seeded generator
(miniasm_amd/bin/pafgen) writes the PAF, the UNMODIFIED reference binary (oracle/_ref/miniasm_ref, built from /root/reference by
oracle/Makefile; the pipeline of /root/reference/main.c:108-199) turns it into GFA once, and what is stored is

    pafgen arguments, md5 + size of the PAF text, RAW md5 + size of the reference's GFA (no normalisation), the reference's wall time.

The reference needs 30 s .. 10 minutes and up to 50 GB of RAM per entry (configs[4]: 500 M overlaps, 30 GB of text) -- that cost stays
here; `tests/test_gpu_cli.py::test_cli_digest_of_the_largest_configurations` and `bench.py`'s `legs.cfg5` only regenerate the PAF,
check ITS md5 (the generator is seeded, the image is the same) and compare the product's GFA digest with the one recorded here."""
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "big.json")
PAFGEN = os.path.join(ROOT, "miniasm_amd", "bin", "pafgen")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")

# name -> generator arguments.  cfg2 / cfg4 / cfg5 are BASELINE.json configs[1] / [3] / [4] as SURVEY.md 8(d) makes them concrete; cfg3 is the
# stand-in for configs[2] (the C. elegans data is not on disk); graph = the fixed-length, graph-heavy 100 M-overlap input (200 M arcs);
# noisy50 = 50 M overlaps with drop-out and false overlaps (both tie walks, the cleaners at scale)
CONFIGS = {
    "cfg2": dict(reads=200000, lines=10000000, seed=1, extra=[]),
    "cfg3": dict(reads=1200000, lines=40000000, seed=4, extra=[]),
    "noisy50": dict(reads=1000000, lines=50000000, seed=3, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    "cfg4": dict(reads=2000000, lines=100000000, seed=2, extra=[]),
    "graph": dict(reads=2000000, lines=100000000, seed=4, extra=["-L", "fixed"]),
    "cfg5": dict(reads=5000000, lines=500000000, seed=3, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    # what a real overlapper writes (pafgen -j / -b / -t, round 6): every coordinate jittered on its own, a tenth of the pairs listed from both sides, the lines grouped by
    # TARGET -- equal sort keys by chance everywhere, no runs of a query's records: the inputs on which the tie path and the record sort are the common path
    "real10": dict(reads=250000, lines=10000000, seed=6, extra=["-j", "30", "-b", "0.1", "-t", "-L", "uniform", "-d", "0.2", "-x", "0.03"]),
    "real50": dict(reads=1000000, lines=50000000, seed=7, extra=["-j", "30", "-b", "0.1", "-t", "-d", "0.2", "-x", "0.03"]),
}


def md5_file(path):
    h, n = hashlib.md5(), 0
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
            n += len(blk)
    return h.hexdigest(), n


def head_tail_md5(path, span=16 << 20):
    """md5 over the first and the last `span` bytes + the size: what the GPU-side checks use to recognise the regenerated PAF without hashing 30 GB"""
    n = os.path.getsize(path)
    h = hashlib.md5()
    with open(path, "rb") as f:
        h.update(f.read(min(span, n)))
        if n > span:
            f.seek(max(span, n - span))
            h.update(f.read())
    h.update(str(n).encode())
    return h.hexdigest()


def md5_stdout(cmd):
    """run cmd, digest its stdout while it runs; returns (md5, bytes, wall seconds, max RSS in MB)"""
    h, n, t0 = hashlib.md5(), 0, time.time()
    with subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) as pr:
        for blk in iter(lambda: pr.stdout.read(1 << 24), b""):
            h.update(blk)
            n += len(blk)
        _, rc, ru = os.wait4(pr.pid, 0)
        pr.returncode = os.waitstatus_to_exitcode(rc)
    if pr.returncode != 0:
        raise RuntimeError("%s: exit %d" % (cmd[0], pr.returncode))
    return h.hexdigest(), n, time.time() - t0, ru.ru_maxrss // 1024


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(CONFIGS)
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/miniasm_ref is missing: run `make -C oracle ref` where /root/reference exists")
    gold = json.load(open(OUT)) if os.path.exists(OUT) else {"reference_version": "0.3-r179", "what": __doc__.split("\n\n")[2], "inputs": {}}
    tmp = os.environ.get("MA_BIG_DIR", "/tmp")
    for name in names:
        cfg = CONFIGS[name]
        have = gold["inputs"].get(name)
        if have and have["pafgen"] == cfg and "--force" not in sys.argv and "paf_head_tail_md5" in have:
            print("[make_big] %s: already recorded" % name)
            continue
        paf = os.path.join(tmp, "make_big_%s.paf" % name)
        t0 = time.time()
        subprocess.run([PAFGEN, "-r", str(cfg["reads"]), "-n", str(cfg["lines"]), "-s", str(cfg["seed"])] + cfg["extra"] + ["-o", paf], check=True, stderr=subprocess.DEVNULL)
        t_gen = time.time() - t0
        try:
            paf_md5, paf_bytes = md5_file(paf)
            ht = head_tail_md5(paf)
            if have and have["pafgen"] == cfg and "--force" not in sys.argv:  # recorded before the head/tail digest existed: the text must be the recorded one, the reference need not run again
                assert (have["paf_md5"], have["paf_bytes"]) == (paf_md5, paf_bytes), "%s: the generator no longer writes the recorded text" % name
                have["paf_head_tail_md5"] = ht
                with open(OUT, "w") as f:
                    json.dump(gold, f, indent=1, sort_keys=True)
                print("[make_big] %s: head/tail digest added" % name)
                continue
            gfa_md5, gfa_bytes, wall, rss = md5_stdout(["taskset", "-c", "0", REF_BIN, paf])
        finally:
            os.remove(paf)
        gold["inputs"][name] = dict(pafgen=cfg, paf_md5=paf_md5, paf_bytes=paf_bytes, paf_head_tail_md5=ht, gfa_md5=gfa_md5, gfa_bytes=gfa_bytes,
                                    reference_wall_s=round(wall, 1), reference_max_rss_mb=rss, pafgen_s=round(t_gen, 1),
                                    host="%s, %d cores" % (next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"), os.cpu_count()))
        with open(OUT, "w") as f:
            json.dump(gold, f, indent=1, sort_keys=True)
        print("[make_big] %s: paf %s (%d B), gfa %s (%d B), reference %.1f s, %d MB" % (name, paf_md5, paf_bytes, gfa_md5, gfa_bytes, wall, rss), flush=True)


if __name__ == "__main__":
    main()
