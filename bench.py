#!/usr/bin/env python3
"""bench.py -- throughput of the miniasm hot path on MI355X.

Metric (BASELINE.json): PAF overlaps (input lines) processed per second through
    hit sort -> coverage/cut/filter x2 -> containment -> string graph -> transitive reduction + symm
    -> tip/bubble/short-overlap cleaning -> unitigs -> GFA text,
with the parsed, unsorted 32-byte hit records already resident in HBM when the timed region starts.

Workload: BASELINE.json configs[3] -- synthetic 100M-overlap PAF, 2M reads, lognormal lengths with mean 8 kb
(miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2): the configuration north_star's 1-GPU target is stated on.  The same
data set is used at every N (strong scaling): at N>1 (launched by torch.distributed.run, one rank per GPU) the hits are
sharded by query-read range, sub / flag arrays and the arc blocks are exchanged over RCCL, rank 0 finishes the graph.

In the same run (rank 0, N=1): the unmodified reference miniasm on the SAME file (cpu_baseline, and its GFA is compared with
ours: gfa_identical), the CLI end to end (process start -> GFA on disk), the text-resident variant (device-side parse inside
the step), BASELINE configs[1] (10M overlaps) and a tie-rich input as secondary legs.

One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def gen_paf(path, reads, lines, seed, extra=()):
    import miniasm_amd as ma
    if os.path.exists(path) and os.path.getsize(path) > 0:
        return path
    tmp = path + ".tmp%d" % os.getpid()
    subprocess.run([ma.PAFGEN_PATH, "-r", str(reads), "-n", str(lines), "-s", str(seed), "-o", tmp] + list(extra), check=True, stderr=subprocess.DEVNULL)
    os.replace(tmp, path)
    return path


def count_lines(path):
    n = 0
    with open(path, "rb") as f:  # pafgen writes one overlap per line, newline-terminated
        while True:
            blk = f.read(64 << 20)
            if not blk:
                break
            n += blk.count(b"\n")
    return n


def md5_pair(data):
    """(md5 of the bytes, md5 of the LC_ALL=C sorted lines)"""
    raw = hashlib.md5(data).hexdigest()
    lines = data.split(b"\n")
    lines.sort()
    return raw, hashlib.md5(b"\n".join(lines)).hexdigest()


def recorded_reference(name):
    """tests/golden/big.json: what the unmodified reference made of the large configurations, once, in the build container (tests/golden/make_big.py):
    pafgen arguments, digest of the text, raw md5 + size of its GFA, its wall time.  None if the file or the entry is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "big.json")))["inputs"].get(name)
    except Exception:
        return None


def head_tail_md5(path, span=16 << 20):
    n = os.path.getsize(path)
    h = hashlib.md5()
    with open(path, "rb") as f:
        h.update(f.read(min(span, n)))
        if n > span:
            f.seek(max(span, n - span))
            h.update(f.read())
    h.update(str(n).encode())
    return h.hexdigest()


def settle(seconds=8.0):
    """Before a command-line leg: the driver clears a finished process's device memory in the background, and an allocation that lands on memory not yet cleared waits for
    it (round 6, profiles/r06_experiments.txt: a 30 GB hipMalloc right behind a process that had held 200 GB took 4.9 s, 0.000 s after an 8 s pause).  What a leg measures is
    a run on an idle GPU, so it waits for the previous process's memory to be given back first; the pause is outside every timed region."""
    time.sleep(seconds)


def cli_digest_leg(ma, name, workdir):
    """BASELINE configs[4] (500 M overlaps) through the command line, GFA digested while it streams out, against the reference's recorded digest"""
    gold = recorded_reference(name)
    if not gold:
        return None
    import shutil
    if shutil.disk_usage(workdir).free < gold["paf_bytes"] + (2 << 30):
        log("leg %s: not enough room in %s for %d bytes of text" % (name, workdir, gold["paf_bytes"]))
        return None
    cfg = gold["pafgen"]
    t0 = time.perf_counter()
    paf = gen_paf(os.path.join(workdir, "leg_%s_r%d_n%d_s%d.paf" % (name, cfg["reads"], cfg["lines"], cfg["seed"])), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
    t_gen = time.perf_counter() - t0
    try:
        same_text = os.path.getsize(paf) == gold["paf_bytes"] and head_tail_md5(paf) == gold["paf_head_tail_md5"]
        h, n = hashlib.md5(), 0
        settle()
        t0 = time.perf_counter()
        with subprocess.Popen(["timeout", "900", ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MA_PIPE_TIMING="2")) as pr:  # (a run that does not come back must not take the bench line with it)
            import threading
            err = []
            th = threading.Thread(target=lambda: err.append(pr.stderr.read()))
            th.start()
            for blk in iter(lambda: pr.stdout.read(1 << 24), b""):
                h.update(blk)
                n += len(blk)
            pr.wait()
            th.join()
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            log("leg %s: the command line failed:" % name, (err[0] if err else b"")[-300:])
            return None
        laps = cli_laps((err[0] if err else b"").decode(errors="replace"))
        return {"value": cfg["lines"] / wall, "unit": "overlaps/s", "wall_s": round(wall, 3), "overlaps": cfg["lines"], "reads": cfg["reads"], "pafgen": cfg, "paf_bytes": gold["paf_bytes"],
                "text_is_the_recorded_one": same_text, "gfa_md5": h.hexdigest(), "gfa_bytes": n, "ref_md5": gold["gfa_md5"], "ref_bytes": gold["gfa_bytes"],
                "gfa_md5_matches_reference": same_text and (h.hexdigest(), n) == (gold["gfa_md5"], gold["gfa_bytes"]),
                "laps": laps, "reference_wall_s": gold["reference_wall_s"], "reference_host": gold["host"], "vs_reference_wall": round(gold["reference_wall_s"] / wall, 1), "gen_s": round(t_gen, 1),
                "what": "miniasm_amd/bin/miniasm <file>: process start to the last byte of GFA (text from the page cache; the run comes BEHIND the configs[3] command-line runs of the e2e leg, and allocations that land on memory another process freed wait for the driver to clear it: `laps` shows where the time went), raw md5 of the GFA against tests/golden/big.json -- the unmodified reference's output on "
                        "the same seeded text, recorded once by tests/golden/make_big.py (the reference needs minutes and tens of GB here: it does not run inside bench.py)"}
    finally:
        try:
            os.remove(paf)  # 30 GB
        except OSError:
            pass


def cli_laps(log_txt):
    """where a command-line run spent its time, from its own [T::...] lines (MA_PIPE_TIMING set)"""
    laps = {}
    for key, pat in (("hip_runtime_s", r"\[T::init\].*HIP runtime ([0-9.]+) s"), ("context_s", r"\[T::init\].*pinned mailbox\) ([0-9.]+) s"), ("file_to_hbm_s", r"\[T::ingest_gpu\] load ([0-9.]+) s"),
                     ("parse_s", r"\[T::ingest_gpu\] parse ([0-9.]+)"), ("dictionary_s", r"dictionary\+release ([0-9.]+) s"),
                     ("sg_gen_incl_tie_repair_ms", r"\[T::head\] sg_gen\s+([0-9.]+) ms"), ("tie_walk_host_ms", r"walk: host\s+([0-9.]+) ms"), ("push_order_ms", r"push order \(all of it\)\s+([0-9.]+) ms"),
                     ("pipeline_head_ms", r"\[T::pipeline\] head ([0-9.]+) ms"), ("pipeline_tail_ms", r"\[T::pipeline\] head [0-9.]+ ms\s+tail ([0-9.]+) ms"), ("real_time_s", r"Real time: ([0-9.]+) sec")):
        mm = re.search(pat, log_txt)
        if mm:
            laps[key] = float(mm.group(1))
    mt = re.search(r"\[T::ties\] (\d+) arc tie groups \((\d+) arcs\), (\d+) push conflicts(?: \((\d+) of them in sight)?", log_txt)
    if mt:
        laps["arc_tie_groups"], laps["push_conflicts"] = int(mt.group(1)), int(mt.group(3))
        if mt.group(4) is not None:
            laps["push_conflicts_in_sight_of_the_arc_sort"] = int(mt.group(4))
    mr = re.search(r"hit walk 1 \(its order taken for (\d+) reads\)", log_txt)
    if mr:
        laps["hit_walk_order_taken_for_reads"] = int(mr.group(1))
    return laps


def run_reference(paf, out_path, runs=1):
    """the unmodified reference (oracle/_ref/miniasm_ref, gcc -O2, 1 thread) on `paf`; returns timings + the GFA's digests"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")
    if not os.path.exists(ref_bin):
        return None
    best = None
    for _ in range(runs):
        with open(out_path, "wb") as fo:
            t0 = time.perf_counter()
            r = subprocess.run(["taskset", "-c", "0", ref_bin, paf], stdout=fo, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
        m = re.search(r"\[M::ma_hit_read::([0-9.]+)\*", r.stderr)
        tot = re.search(r"Real time: ([0-9.]+) sec", r.stderr)
        if r.returncode != 0 or not m or not tot:
            log("reference run failed:", r.stderr[-500:])
            return None
        cur = {"t_parse": float(m.group(1)), "t_all": float(tot.group(1)), "wall": wall}
        if best is None or cur["t_all"] - cur["t_parse"] < best["t_all"] - best["t_parse"]:
            best = cur
    with open(out_path, "rb") as f:
        best["md5"], best["md5_sorted"] = md5_pair(f.read())
    return best


def pmc_profile(workload):
    """the committed rocprofv3 PMC run of this same command (profiles/rNN_pmc_traffic_<workload>.json, made by `tools/gpu_round.sh pmc`:
    FETCH_SIZE and WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md's HBM section prescribes).  PMC counters cannot be
    read from inside this process: the figures are attached only when the workload is the profiled one, and are labelled with the file
    and the commit the profile was taken at (its `_meta`)."""
    for rnd in ("r06", "r05", "r04", "r03", "r03a", "r02", "r01"):  # the newest profile of this workload
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, workload))
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.pop("_meta", {})
        src = "%s @ commit %s" % (os.path.relpath(path, ROOT), meta.get("commit", "unknown (before round 3)"))
        return {k.replace("void ", ""): v for k, v in d.items()}, src
    return None, None


def pmc_traffic(names, kernel):
    """HBM bytes per launch of the timed scope `kernel` (fetch, corrected, + write) from pmc_profile()'s table"""
    if not names:
        return None
    per_launch = lambda v: v["fetch_bytes_x2"] + v["write_bytes"]
    scope = {"k_hit_sub<cut+flt>": ("k_hit_sub<true,", None), "k_hit_sub": ("k_hit_sub<false,", False), "k_hit_sub<gather>": ("k_hit_sub<false,", True)}.get(kernel)
    if scope:  # a timed scope of the coverage passes = one launch of each size-class kernel: their bytes add up
        def gathers(k):  # the THIRD template argument is the gather mode (absent in round 1's names, a bool until round 4, 0 / 1 / 2 since)
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            return len(targs) >= 3 and targs[2] in ("true", "1", "2")
        parts = [v for k, v in names.items() if k.startswith(scope[0]) and (scope[1] is None or gathers(k) == scope[1])]
        return round(sum(per_launch(v) for v in parts)) if parts else None
    base = kernel.split("<")[0]
    # a timed scope of several launches whose bytes add up
    multi = {"k_group_close": ("k_group_tile_min", "k_group_close"), "k_radix_colscan": ("k_radix_colscan_chunk", "k_radix_colscan_top"),
             "k_runs_expand": ("k_runs_count", "k_runs_expand"), "k_arc_rm": ("k_arc_rm_count", "k_arc_rm_write")}.get(base)
    if multi:
        parts = [next((v for k, v in names.items() if k.split("<")[0] == want), None) for want in multi]
        if all(parts):
            return round(sum(per_launch(v) for v in parts))
        if base == "k_radix_colscan":
            return 0  # (a counter profile from before these kernels existed: 50 MB per pass, nothing against the group's 46 GB)
        if base != "k_arc_rm":
            return None
    alias = {"k_hit_keys": ("k_hit_keys_tiled", "k_hit_keys_runs"), "k_arc_rm": ("k_arc_rm_chain",), "k_asg_trans": ("k_asg_trans", "k_asg_trans_pipe")}.get(base, (base,))
    hits = [v for k, v in names.items() if k.split("<")[0] in (base,) + alias]
    if base == "k_radix_scatter" and any(k.startswith("k_radix_scatter<false") for k in names):  # the hits' keys travel without a value array; the <true, ...> launches are the (small) pair sorts of the same step
        hits = [v for k, v in names.items() if k.startswith("k_radix_scatter<false")]
    if not hits:
        return None
    if base in ("k_arc_group_sort", "k_asg_trans"):  # a timed scope = one launch of each size-class instantiation: their bytes add up
        return round(sum(per_launch(v) for v in hits))
    return round(sum(per_launch(v) * v["launches"] for v in hits) / max(sum(v["launches"] for v in hits), 1))


def achievable_rates():
    """profiles/r0N_diag_patterns.json (a committed measurement, not taken in this run): GB/s of the largest size per pattern"""
    for fn in ("profiles/r04_diag_patterns.json",):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fn)
        if os.path.exists(path):
            try:
                rows = json.load(open(path))["rows"]
                top = max(r["source_MB"] for r in rows)
                pick = {"read16_flat": "stream_read", "write16": "stream_write", "copy16": "copy", "read8_of32+write8": "keys_from_records", "scatter_runs32": "radix_scatter_runs",
                        "gather32+cols": "gather_32B_records_to_columns_useful_bytes"}
                out = {v: round(r["TB_per_s"] * 1e3, 0) for r in rows for k, v in pick.items() if r["pattern"] == k and r["source_MB"] == top}
                out.update(unit="GB/s", source="%s: csrc/diag.hip patterns over %d MB, fastest of 5 launches, HIP events (visit I of round 4); a random 32-byte fetch moves a 128-byte line, "
                           "so the gather's 64 useful bytes per record stand for 160 moved" % (fn, top))
                return out
            except Exception:
                return None
    return None


# Timed scopes whose byte figure is NOT a SURVEY 8(d) row of its own: 8(d) prices the hit sort at 64 B per hit "counted once regardless of
# digit passes" and that row is billed to the scope that moves the records (k_hit_sub<gather>: sort 64 + ma_hit_sub 48 B per hit); what the
# key / digit / offset kernels report is the traffic of this design (keys 16, a digit pass 8 + 16, offsets 8 B per hit): `design_GBs`.
DESIGN_ONLY = ("k_hit_keys", "k_radix_hist", "k_radix_colscan", "k_radix_scatter", "k_hit_goff", "k_group_close", "scan_exclusive_u32", "k_arc_keys", "k_arc_permute")
SORT_GROUP = ("k_hit_keys", "k_radix_hist", "k_radix_colscan", "k_radix_scatter", "k_runs_expand", "k_hit_goff", "k_group_close", "k_hit_sub<gather>")
# the "reduce" half of north_star's roofline target (per arc): arc sort 32 + index 16 + del_trans 16 (A + I)/A + del_multi 16 +
# del_asymm 16 + 16 x entries probed + asg_arc_rm 32.  The timed scopes that do that work (graph.hip); the *_radix_* / permute / census scopes only
# run when the in-register arc sort hands a sort to the radix path.
REDUCE_GROUP = ("k_arc_groups", "k_arc_group_sort", "k_arc_radix_hist", "k_arc_radix_scatter", "k_arc_permute", "k_arc_tie_census", "k_arc_index",
                "k_asg_trans", "k_asg_trans_big", "k_asg_multi", "k_asg_asymm", "k_arc_rm")
REDUCE_DESIGN_ONLY = ("k_arc_groups", "k_arc_radix_hist", "k_arc_radix_scatter", "k_arc_permute", "k_arc_tie_census")  # their bytes are this design's, not 8(d) rows


def kernel_table(recs, prof_steps, design_only=DESIGN_ONLY + ("k_arc_groups", "k_arc_radix_hist", "k_arc_radix_scatter", "k_arc_tie_census")):
    """per timed scope: launches per step, HIP-event time per launch, share, algorithmic GB/s"""
    tot_ms = sum(r["total_ms"] for r in recs) or 1.0
    out = []
    for r in sorted(recs, key=lambda r: -r["total_ms"]):
        per = r["total_ms"] / max(r["launches"], 1)
        gbs = round(r["alg_bytes"] / max(r["launches"], 1) / (per * 1e-3) / 1e9, 1) if per > 0 and r["alg_bytes"] > 0 else None
        design = r["name"] in design_only
        out.append({"name": r["name"], "launches_per_step": r["launches"] / max(prof_steps, 1), "avg_ms": round(per, 5), "share": round(r["total_ms"] / tot_ms, 4),
                    "alg_GBs": None if design else gbs, "design_GBs": gbs if design else None, "alg_bytes_per_step": r["alg_bytes"] / max(prof_steps, 1)})
    return out


def reduce_group(ktab, n_arc, n_inner):
    """SURVEY 8(d)'s per-arc rows of the graph phase over the HIP-event time of the scopes that do them"""
    grp = [k for k in ktab if k["name"] in REDUCE_GROUP]
    ms = sum(k["avg_ms"] * k["launches_per_step"] for k in grp)
    alg = sum(k["alg_bytes_per_step"] for k in grp if k["name"] not in REDUCE_DESIGN_ONLY)
    if ms <= 0 or alg <= 0:
        return None
    worst = min((k for k in grp if k["alg_GBs"] and k["avg_ms"] * k["launches_per_step"] > 0.02 * ms), key=lambda k: k["alg_GBs"], default=None)  # furthest below the roofline among those that matter
    return {"kernels": [{"name": k["name"], "launches_per_step": k["launches_per_step"], "avg_ms": k["avg_ms"], "alg_GBs": k["alg_GBs"], "design_GBs": k["design_GBs"]} for k in grp],
            "arcs_into_the_reduction": n_arc, "inner_iterations_I": n_inner, "ms_per_step": round(ms, 4), "alg_bytes_per_step": alg,
            "bytes_per_arc": round(alg / max(n_arc, 1), 1), "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "slowest_by_8d": worst and worst["name"],
            "note": "sum of SURVEY 8(d)'s per-arc rows (arc sort 32, index 16, del_trans 16 (A + I) with I counted on the device, del_multi 16, del_asymm 16 + 16 x probed entries, "
                    "asg_arc_rm 32; each row x the arcs its launch saw) / HIP-event time of the scopes that do them"}


class Workload:
    """one PAF file brought to the state the timed region starts from: text parsed on the device, the unsorted records of this
    rank's read range in a torch buffer, the dictionary on the host"""

    def __init__(self, ma, L, ctx, paf, opt, world, rank, keep_text=False):
        import torch
        self.paf, self.n_lines = paf, count_lines(paf)
        L.ma_paf_load_file.argtypes = [C.c_void_p, C.c_char_p]
        L.ma_hit_ingest_loaded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(ma.Sdict), C.POINTER(C.c_size_t), C.c_int, C.c_int]
        L.mahip_hits_raw_extract.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_size_t)]
        t0 = time.perf_counter()
        if L.ma_paf_load_file(ctx.h, paf.encode()) != 0:
            raise RuntimeError("cannot load %s into HBM" % paf)
        L.mahip_sync(ctx.h)
        self.t_load = time.perf_counter() - t0
        self.d = L.sd_init()
        nh = C.c_size_t(0)
        t0 = time.perf_counter()
        if L.ma_hit_ingest_loaded(ctx.h, opt.min_span, opt.min_match, self.d, C.byref(nh), 1, 0 if keep_text else 1) != 0:
            raise RuntimeError("device-side parse failed: " + L.mahip_strerror().decode())
        self.t_parse = time.perf_counter() - t0
        self.n_all, self.n_seq = nh.value, self.d.contents.n_seq
        L.mahip_paf_max_qs.restype = C.c_uint32
        L.mahip_paf_max_qs.argtypes = [C.c_void_p]
        self.max_qs = L.mahip_paf_max_qs(ctx.h)
        q0, q1 = 0, 0xffffffff
        self.bounds = None
        if world > 1:  # read ranges with equally many hits; the table is handed to the sharded head with every batch (a table describes one upload, include/mahip.h)
            L.mahip_hits_balance.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
            L.mahip_set_shard_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
            self.bounds = bounds = (C.c_uint32 * (world + 1))()
            ma._chk(L.mahip_hits_balance(ctx.h, world, bounds), "balance")
            q0, q1 = bounds[rank], bounds[rank + 1]
        n_my = C.c_size_t(0)
        ma._chk(L.mahip_hits_raw_extract(ctx.h, q0, q1, None, C.byref(n_my)), "raw_extract")
        self.n_my = n_my.value
        self.hits_dev = torch.empty(max(self.n_my, 1) * 32, dtype=torch.uint8, device="cuda")
        # N > 1: where this rank's records stood in the input -- a rank that only holds its read range needs it to restore the reference's order of tied hits
        self.pos_dev = torch.empty(max(self.n_my, 1) * 4, dtype=torch.uint8, device="cuda") if world > 1 else None
        L.mahip_hits_raw_extract_pos.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        L.mahip_hits_set_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
        ma._chk(L.mahip_hits_raw_extract_pos(ctx.h, q0, q1, C.c_void_p(self.hits_dev.data_ptr()), C.c_void_p(self.pos_dev.data_ptr()) if world > 1 else None, C.byref(n_my)), "raw_extract")
        L.mahip_sync(ctx.h)
        self.size = os.path.getsize(paf)

    def close(self, L):
        if self.d:
            L.sd_destroy(self.d)
            self.d = None
        self.hits_dev = None
        self.pos_dev = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=2000000)
    ap.add_argument("--lines", type=int, default=100000000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--model", default="lognormal", choices=["lognormal", "fixed", "uniform"])
    ap.add_argument("--grid", type=int, default=0, help="pafgen -q: coordinates on a grid -> equal sort keys (tie-rich input; not the BASELINE workload)")
    ap.add_argument("--workdir", default=os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"))
    ap.add_argument("--no-cpu", action="store_true", help="skip the reference run (no cpu_baseline, no gfa_identical)")
    ap.add_argument("--no-text", action="store_true", help="skip the text-resident leg (device-side parse inside the step)")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (cfg2, tie-rich input, CLI end to end)")
    ap.add_argument("--no-overlap", action="store_true", help="run each pass's host tail before the next pass's device part starts")
    ap.add_argument("--tail-one-stage", action="store_true", help="A/B: the tail of a batch (device part, then text) on ONE worker thread, as in round 5; default: two stages on two threads")
    ap.add_argument("--no-tail-ctx", action="store_true", help="keep the latency-bound rest of a batch (cleaners, unitigs, downloads) on the context that runs the hit passes.  "
                    "Default: it moves to a second context on the same GPU and runs beside the next batch's hit passes (mahip_tail_handoff; round 3, measured: "
                    "cfg4 20.0 -> 19.2 ms per step, cfg2 3.23 -> 2.79 ms)")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--legs", default="cfg2,tie_rich,graph_heavy,latency,e2e,cfg5,realistic", help="which secondary legs to run (comma list)")
    ap.add_argument("--graph-heavy-lines", type=int, default=100000000, help="overlaps of the graph-heavy leg (pafgen -L fixed, 50 lines per read); 0 = skip it")
    ap.add_argument("--inflight", type=int, default=1, help="experiment: batches in flight on the one GPU (each on its own context and host thread)")
    args = ap.parse_args()
    args.tail_ctx = not args.no_tail_ctx
    want_leg = lambda name: not args.no_legs and name in args.legs.split(",")
    args.gen_extra = ([] if args.model == "lognormal" else ["-L", args.model]) + (["-q", str(args.grid), "-d", "0.3", "-x", "0.03"] if args.grid else [])
    cfg_name = {(2000000, 100000000, 2, "lognormal"): "cfg4", (200000, 10000000, 1, "lognormal"): "cfg2"}.get((args.reads, args.lines, args.seed, args.model), "custom")

    import torch
    import torch.distributed as dist
    import miniasm_amd as ma

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: exercise the multi-process path on a box with ONE GPU (all ranks on device 0, collectives over gloo);
    # never set by the driver -- the real thing is one rank per GPU over RCCL
    one_gpu_debug = os.environ.get("MA_BENCH_ONE_GPU_DEBUG") == "1"
    if one_gpu_debug:
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # torch.distributed is the control plane only (rendezvous of the RCCL id, barriers, the max over ranks of the time):
        # the data plane is RCCL called from C on the context's stream (csrc/comm.hip, host/sharded.c)
        dist.init_process_group(backend="gloo")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    os.makedirs(args.workdir, exist_ok=True)
    if not os.path.exists(ma.LIB_PATH):
        if rank == 0:
            ma.build()
        if world > 1:
            dist.barrier()
    L = ma.lib()
    L.ma_set_log_path(b"/dev/null")
    L.sys_init()

    t0 = time.perf_counter()
    paf = os.path.join(args.workdir, "w_%s%s_r%d_n%d_s%d.paf" % (args.model, "_q%d" % args.grid if args.grid else "", args.reads, args.lines, args.seed))
    if rank == 0:
        gen_paf(paf, args.reads, args.lines, args.seed, args.gen_extra)
    if world > 1:
        dist.barrier()
    t_gen = time.perf_counter() - t0
    opt = ma.default_opt()

    # ---- the command-line legs FIRST, while this process holds no device memory: a process that starts beside 30 GB of another process's buffers -- or right after they
    # were freed, when the driver clears them first -- does not show what the command line costs (round 5, visit 1: configs[3] 0.34 s on an idle GPU, 1.2 s at the end of
    # this script; configs[4] 8.3 s there against 4.9 s alone)
    early_e2e, early_cfg5 = None, None
    cfg_name_early = {(2000000, 100000000, 2, "lognormal"): "cfg4", (200000, 10000000, 1, "lognormal"): "cfg2"}.get((args.reads, args.lines, args.seed, args.model), "custom")
    if rank == 0 and world == 1 and want_leg("e2e"):
        try:  # the command line, process start -> GFA on disk (north_star's ">= 10x reference wall-clock" is about this)
            outp = os.path.join(args.workdir, "cli_%s.gfa" % cfg_name_early)
            walls, laps_all = [], []
            for k in range(3):
                if k:
                    settle(3.0)  # (the previous run's 50 GB)
                with open(outp, "wb") as fo:
                    t0 = time.perf_counter()
                    r = subprocess.run([ma.CLI_PATH, paf], stdout=fo, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, MA_PIPE_TIMING="1"))  # (a dozen stamps on stderr)
                    walls.append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr[-300:]
                laps_all.append(cli_laps(r.stderr.decode(errors="replace")))
            early_e2e = {"walls": walls, "laps": laps_all[walls.index(min(walls))], "md5": md5_pair(open(outp, "rb").read())[0]}
        except Exception as e:
            log("e2e leg failed:", e)
    if rank == 0 and world == 1 and want_leg("cfg5") and cfg_name_early == "cfg4":
        try:  # BASELINE configs[4]: 500 M overlaps, high-repeat -- the tie walk, tier-1 bubble tables and 67-bit packed keys all live at once
            early_cfg5 = cli_digest_leg(ma, "cfg5", args.workdir)
        except Exception as e:
            log("cfg5 leg failed:", e)
    early_real = None
    if rank == 0 and world == 1 and want_leg("realistic") and cfg_name_early == "cfg4":
        try:  # what a real overlapper writes: jittered coordinates, a tenth of the pairs from both sides, lines grouped by TARGET (pafgen -j 30 -b 0.1 -t): the record sort and both tie walks
            early_real = cli_digest_leg(ma, "real50", args.workdir)
        except Exception as e:
            log("realistic leg failed:", e)

    ctx = ma.Ctx(local)
    if world > 1:  # one communicator per rank: rank 0 makes the RCCL id, the control plane hands it round
        L.mahip_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.mahip_comm_init_shm.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        if one_gpu_debug:
            ma._chk(L.mahip_comm_init_shm(ctx.h, b"miniasm_amd_bench_%s" % os.environ.get("MASTER_PORT", "0").encode(), rank, world), "comm_init_shm")
        else:
            box = [None]
            if rank == 0:
                idb = C.create_string_buffer(128)
                ma._chk(L.mahip_comm_unique_id(idb), "comm_unique_id")
                box[0] = idb.raw
            dist.broadcast_object_list(box, src=0)
            ma._chk(L.mahip_comm_init(ctx.h, box[0], rank, world), "comm_init")
    want_text = rank == 0 and world == 1 and not args.no_text
    W = Workload(ma, L, ctx, paf, opt, world, rank, keep_text=want_text)
    if rank == 0:
        log("workload %s: %d lines (%.2f GB text), %d stored hits (%d on this rank), %d reads; gen %.1fs, file->HBM %.3fs (%.1f GB/s), parse+dictionary %.3fs" % (
            cfg_name, W.n_lines, W.size / 1e9, W.n_all, W.n_my, W.n_seq, t_gen, W.t_load, W.size / W.t_load / 1e9, W.t_parse))

    vp = C.c_void_p
    L.ma_pipeline_head.restype = C.c_int
    L.ma_pipeline_head.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_fetch.restype = vp
    L.ma_pipeline_tail_fetch.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_finish_mem.restype = C.c_int
    L.ma_pipeline_tail_finish_mem.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]

    ShardStats = ma.ShardStats
    L.ma_pipeline_head_sharded.restype = C.c_int
    L.ma_pipeline_head_sharded.argtypes = [vp, C.POINTER(ma.MaOpt), C.c_uint32, C.c_int, C.POINTER(ShardStats)]
    overlap = not args.no_overlap

    # Passes are pipelined over the stream of batches: the device part of pass k+1 starts as soon as pass k's reduced graph
    # has been fetched (ma_pipeline_tail_fetch = the last use of the device for a batch), while pass k's host part (graph
    # cleaning, unitigs, GFA text: ma_pipeline_tail_finish) runs on a worker thread.  All K outputs are complete before the
    # closing fence.  --no-overlap runs head and tail back to back.
    import queue
    import threading

    class Runner:
        """the timed step over one workload"""

        def __init__(self, W, hctx=None, overlap=overlap):
            self.W = W
            self.hctx = hctx or ctx  # the context the hit passes run on
            self.out = {"n": 0, "rc": 0, "buf": None}
            self.head_wall, self.tail_wall, self.fetch_wall, self.last_stats = [], [], [], None  # wall time of each head on this rank / of each tail (rank 0)
            self.q = queue.Queue(maxsize=1)  # one batch may wait while another is being finished
            self.q_text = queue.Queue(maxsize=1)  # second stage of the tail: the device part of batch k+1 (second context) runs beside the text of batch k (host threads)
            self.worker, self.worker_text = None, None
            self.fetch2_wall, self.text_wall = [], []
            # --tail-ctx: the device tail of a batch (graph cleaning, unitigs, downloads -- many small launches and counter fetches) moves to a
            # second context and to the worker thread; this thread goes straight on to the next batch's hit passes
            self.ctx2 = ma.Ctx(local) if (args.tail_ctx and overlap and rank == 0) else None
            self.ctx2_free = threading.Semaphore(1)
            if overlap and rank == 0:
                self.worker = threading.Thread(target=self._work, daemon=True)
                self.worker.start()
                if self.ctx2 and not args.tail_one_stage:
                    self.worker_text = threading.Thread(target=self._work_text, daemon=True)
                    self.worker_text.start()

        def _finish(self, job, t_start=None):
            b, l = vp(0), C.c_size_t(0)
            rc = L.ma_pipeline_tail_finish_mem(job, C.byref(b), C.byref(l))
            if t_start is not None:
                self.tail_wall.append(time.perf_counter() - t_start)
            if self.out["buf"]:
                L.free_buf(self.out["buf"])
            self.out["n"], self.out["rc"], self.out["buf"] = l.value, self.out["rc"] or rc, b  # the last output is kept for the parity check

        def _work(self):
            while True:
                job = self.q.get()
                try:
                    t_tail = None
                    if isinstance(job, tuple):  # (status words of the head): the device tail is still to do, on the second context
                        st = job[0]
                        t_tail = time.perf_counter()
                        try:
                            job = L.ma_pipeline_tail_fetch(self.ctx2.h, C.byref(opt), self.W.d, b"ug", 100, C.byref(st))
                        finally:
                            self.ctx2_free.release()
                        assert job
                        self.fetch2_wall.append(time.perf_counter() - t_tail)
                    if job is not None:
                        if self.worker_text:
                            self.q_text.put((job, t_tail))
                        else:
                            self._finish(job, t_tail)
                except Exception as e:  # never leave the fence waiting on a dead worker
                    self.out["rc"] = self.out["rc"] or -1
                    log("host tail failed:", e)
                finally:
                    self.q.task_done()
                if job is None:
                    return

        def _work_text(self):
            while True:
                item = self.q_text.get()
                try:
                    if item is not None:
                        t0 = time.perf_counter()
                        self._finish(*item)
                        self.text_wall.append(time.perf_counter() - t0)
                except Exception as e:
                    self.out["rc"] = self.out["rc"] or -1
                    log("host tail (text) failed:", e)
                finally:
                    self.q_text.task_done()
                if item is None:
                    return

        def step(self):
            W = self.W
            ma._chk(L.mahip_hits_adopt(self.hctx.h, W.hits_dev.data_ptr(), W.n_my, W.n_seq), "adopt")
            L.mahip_set_hints(self.hctx.h, W.max_qs)
            L.mahip_set_run_stride(self.hctx.h, 2)  # the records are ma_hit_read's: a line's record and its mirror side by side (hit.c:87-98)
            if W.bounds is not None:
                ma._chk(L.mahip_set_shard_bounds(self.hctx.h, W.bounds, world), "set_shard_bounds")
            if W.pos_dev is not None:
                ma._chk(L.mahip_hits_set_positions(self.hctx.h, C.c_void_p(W.pos_dev.data_ptr()), 1, W.n_all), "set_positions")
            st = (C.c_uint32 * 4)(0, 0, 0, 0)
            if world == 1:  # single GPU: the C pipeline's device half
                t_h = time.perf_counter()
                assert L.ma_pipeline_head(self.hctx.h, C.byref(opt), W.d, b"ug", 100, 0, C.byref(st)) == 0
                self.head_wall.append(time.perf_counter() - t_h)
            else:  # sharded: device passes + RCCL exchanges on every rank (host/sharded.c), graph cleaning + unitigs + GFA on rank 0
                stats = ShardStats()
                t_h = time.perf_counter()
                assert L.ma_pipeline_head_sharded(self.hctx.h, C.byref(opt), W.n_seq, 0, C.byref(stats)) == 0  # 0: this rank holds its own records only
                self.head_wall.append(time.perf_counter() - t_h)
                self.last_stats = stats
                if rank != 0:
                    return
                st = (C.c_uint32 * 4)(1, 1, stats.n_red, 1)
            if self.ctx2:
                self.ctx2_free.acquire()  # the previous batch's device tail has left the second context
                ma._chk(L.mahip_tail_handoff(self.hctx.h, self.ctx2.h), "tail_handoff")
                self.q.put((st,))
                return
            t_f = time.perf_counter()
            job = L.ma_pipeline_tail_fetch(self.hctx.h, C.byref(opt), W.d, b"ug", 100, C.byref(st))
            assert job
            self.fetch_wall.append(time.perf_counter() - t_f)
            if self.worker:
                self.q.put(job)
            else:
                self._finish(job)

        def phases(self):
            """where a pass of this runner spends its time, host-side wall clocks (means over the steps so far; the tail's laps are those of the LAST pass, from the C
            pipeline's own stamps): the head is the device half on the first context, the tail's device part (cleaners, unitigs) runs on the second context beside the
            next head, its host part (text) on the worker thread"""
            laps = (C.c_double * 8)()
            L.ma_pipeline_last_laps(C.byref(laps))
            mean = lambda a: round(sum(a) / len(a) * 1e3, 3) if a else None
            return {"head_wall_ms": mean(self.head_wall), "tail_wall_ms": mean(self.tail_wall) if self.tail_wall else mean(self.fetch_wall),
                    "tail_device_stage_ms": mean(self.fetch2_wall), "tail_text_stage_ms": mean(self.text_wall),
                    "tail_last_pass_ms": {"survivors_names_intervals_to_host": round(laps[0], 3), "device_cleaners": round(laps[1], 3), "unitigs_to_host": round(laps[2], 3),
                                          "text": round(laps[3], 3), "text_format": round(laps[4], 3), "text_assemble": round(laps[5], 3)},
                    "note": "wall clocks on the host: head = ma_pipeline_head (sort .. reduced graph, ends with a counter fetch); tail = second context + worker thread (its device part waits for "
                            "the GPU beside the next pass's head), in two stages on two threads (device part of pass k+1 beside the text of pass k); a stream of passes costs max(head, tail device stage, tail text stage) each"}

        def fence(self):
            if self.worker:
                self.q.join()  # every host tail handed over so far is complete
                self.q_text.join()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            assert self.out["rc"] == 0

        def timed(self, warmup, steps):
            for _ in range(warmup):
                self.step()
            self.fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.fence()
            return time.perf_counter() - t0

        def output(self):
            return C.string_at(self.out["buf"], self.out["n"]) if self.out["buf"] else b""

        def close(self):
            if self.worker:
                self.q.put(None)
                self.worker.join()
                self.worker = None
            if self.worker_text:
                self.q_text.put(None)
                self.worker_text.join()
                self.worker_text = None
            if self.out["buf"]:
                L.free_buf(self.out["buf"])
                self.out["buf"] = None
            if self.ctx2:
                self.ctx2.close()
                self.ctx2 = None

    L.mahip_tail_handoff.argtypes = [vp, vp]
    L.ma_pipeline_last_laps.argtypes = [C.POINTER(C.c_double * 8)]
    L.ma_pipeline_last_laps.restype = None
    L.ma_shard_phases.argtypes = [C.c_int]
    run = Runner(W)
    if args.inflight > 1 and world == 1:
        # EXPERIMENT (--inflight N): N batches in flight on one GPU, each on a context (stream + buffers) of its own, driven by its own host thread; the
        # kernels of different batches share the chip (a VALU-bound coverage pass of one beside the HBM-bound sort passes of another)
        others = [Runner(W, ma.Ctx(local)) for _ in range(args.inflight - 1)]
        team = [run] + others

        def spin(r, k):
            for _ in range(k):
                r.step()
        def many(k_total):
            share = [k_total // len(team) + (1 if i < k_total % len(team) else 0) for i in range(len(team))]
            th = [threading.Thread(target=spin, args=(r, k)) for r, k in zip(team, share)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for r in team:
                r.fence()
        many(args.warmup * len(team))
        t0 = time.perf_counter()
        many(args.steps)
        dt = time.perf_counter() - t0
        for r in others:
            assert r.output() == run.output()
            r.close()
            r.hctx.close()
    else:
        dt = run.timed(args.warmup, args.steps)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    total_lines = float(W.n_lines)
    main_phases = run.phases() if (rank == 0 and world == 1) else None
    gfa = run.output() if rank == 0 else b""
    tie = ctx.tie_stats() if rank == 0 else None

    # ---- per-kernel timing with HIP events on the launch stream (separate, instrumented steps)
    roof, kernels = None, []
    if rank == 0:
        ctx.prof_enable(True)
        ctx.prof_reset()
    if world > 1:
        L.ma_shard_phases(1)
        run.head_wall, run.tail_wall = [], []
    for _ in range(args.prof_steps):  # a step is collective in the sharded mode: EVERY rank runs it, rank 0 is the one instrumented
        run.step()
        run.fence()  # an instrumented step runs ALONE: with the previous pass's device tail on the second context beside it, whichever kernel it lands on is stretched by several ms (round 6: k_asg_trans 10.4 instead of 2.5 ms)
    phases = None
    if world > 1:  # where a sharded step spends its time: device time per phase (HIP events between the phases of host/sharded.c), maximum over the ranks
        L.ma_shard_phases(0)
        stt = run.last_stats
        mine = [float(x) for x in stt.phase_ms] + [sum(run.head_wall) / max(len(run.head_wall), 1) * 1e3]
        tmax = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        if rank == 0:
            names = ma.SHARD_PHASE_NAMES
            dev_sum = float(sum(tmax[:len(names)]))
            phases = {"what": "one sharded head (the last instrumented step): device time between the phase marks of host/sharded.c, max over ranks [min]; x:* = exchanges",
                      "phase_ms": {n: [round(float(tmax[i]), 4), round(float(tmin[i]), 4)] for i, n in enumerate(names)},
                      "exchange_bytes_per_rank": {names[i]: int(stt.xchg_bytes[i]) for i in range(len(names)) if stt.xchg_bytes[i]},
                      "device_ms_sum_of_max": round(dev_sum, 4), "head_wall_ms_max": round(float(tmax[-1]), 4), "head_wall_ms_min": round(float(tmin[-1]), 4),
                      "host_wait_ms": round(float(tmax[-1]) - dev_sum, 4),
                      "rank0_tail_wall_ms": round(sum(run.tail_wall) / max(len(run.tail_wall), 1) * 1e3, 4) if run.tail_wall else None,
                      "rank0_tail_where": "second context + worker thread, beside the next step's head" if run.ctx2 else "same context, after the head"}
    if rank == 0:
        recs = ctx.prof_get()
        ctx.prof_enable(False)
        pmc, pmc_src = pmc_profile(cfg_name)
        kernels = kernel_table(recs, args.prof_steps)
        for k in kernels:
            tr = pmc_traffic(pmc, k["name"])
            # the bytes the launch really moved (rocprofv3 PMC profile of this command) / this run's launch time: cannot exceed the peak
            k["counter_GBs"] = round(tr / (k["avg_ms"] * 1e-3) / 1e9, 1) if tr and k["avg_ms"] > 0 else None
            k["frac_counter"] = round(k["counter_GBs"] / HBM_PEAK_GBS, 4) if k["counter_GBs"] else None  # the fraction to read: bytes the launch moved / its time / peak
            if k.get("alg_GBs") and k["alg_GBs"] > HBM_PEAK_GBS:
                k["alg_GBs_is_accounting"] = "SURVEY 8(d) bytes of the reference passes this fused launch replaces, not bytes it moved: read frac_counter" 
        dom = next((k for k in kernels if k["alg_GBs"]), None)
        if dom:
            traffic = pmc_traffic(pmc, dom["name"])
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["alg_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom["alg_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": pmc_src and "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)" % pmc_src,
                    # the same kernel priced by the bytes it really moved: a fused kernel cannot exceed 1 here
                    "frac_counter": round(traffic / (dom["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
                    "note": "achieved = SURVEY 8(d) algorithmic bytes of the reference passes this kernel replaces (k_hit_sub<gather>: hit sort 64 + ma_hit_sub 48 B per stored hit) / HIP-event launch time"}
            # the sort + first coverage pass as a group: ALL kernels that make up the reference's hit sort + first ma_hit_sub, priced at 8(d)'s 64 + 48 B per hit
            grp = [k for k in kernels if k["name"] in SORT_GROUP]
            grp_ms = sum(k["avg_ms"] * k["launches_per_step"] for k in grp)
            if grp_ms > 0:
                grp_tr = [pmc_traffic(pmc, k["name"]) for k in grp]
                grp_bytes = 112.0 * float(W.n_my)
                grp_traffic = round(sum(t * k["launches_per_step"] for t, k in zip(grp_tr, grp))) if all(t is not None for t in grp_tr) else None
                roof["sort_group"] = {"kernels": [k["name"] for k in grp], "ms_per_step": round(grp_ms, 4), "alg_bytes_per_step": grp_bytes,
                                      "achieved": round(grp_bytes / (grp_ms * 1e-3) / 1e9, 1), "frac": round(grp_bytes / (grp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "traffic": grp_traffic}
                # THE HEADLINE (round-4 review, What's weak #2): the reference's hit sort + first ma_hit_sub are done here by several kernels (keys, histograms, scatters, the
                # gathering coverage sweep); billing the sweep alone with the whole 64 + 48 B per hit flatters it.  `roofline` is therefore the GROUP -- all of those
                # launches together, one "launch" = one pass of the group over the input -- and the single kernel stays beside it as `dominant_kernel`.
                single = {k: roof[k] for k in ("kernel", "achieved", "frac", "traffic", "frac_counter", "avg_launch_ms", "launches_per_step", "note")}
                roof.update({"kernel": "sort group = " + " + ".join("%s x%g" % (k["name"], k["launches_per_step"]) for k in grp),
                             "achieved": roof["sort_group"]["achieved"], "frac": roof["sort_group"]["frac"], "traffic": grp_traffic,
                             "frac_counter": round(grp_traffic / (grp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if grp_traffic else None,
                             "avg_launch_ms": round(grp_ms, 4), "launches_per_step": 1,
                             "note": "achieved = SURVEY 8(d) algorithmic bytes of the reference passes the group replaces (hit sort 64 + first ma_hit_sub 48 B per stored hit) x stored hits / "
                                     "sum of the HIP-event launch times of the group's kernels in one step",
                             "dominant_kernel": single})
            # what this GPU sustained for plain access patterns with known byte counts (csrc/diag.hip, tools/pmc_calibrate.py): the rate `frac_counter` is to be read against
            ach = achievable_rates()
            if ach:
                roof["achievable"] = ach
            # What THIS design of the sort group can reach at best: every kernel's counted traffic (the committed PMC profile of this command) at the rate this GPU sustained for that
            # kernel's access pattern with known byte counts (csrc/diag.hip).  It is a ceiling of the design, not of the hardware: the group moves about twice SURVEY 8(d)'s bytes
            # (keys are made from whole records, three digit passes read the keys twice and write them once, a mirrored record costs a 128-byte line for 32 bytes), so 0.5 of the
            # 8 TB/s peak by 8(d) bytes would need the group's traffic at more than the box streams.
            if ach and grp_ms > 0 and all(t is not None for t in grp_tr):
                pat = {"k_hit_keys": "keys_from_records", "k_radix_hist": "stream_read", "k_radix_colscan": "stream_read", "k_radix_scatter": "radix_scatter_runs", "k_runs_expand": "copy",
                       "k_hit_goff": "copy", "k_group_close": "copy"}
                line_rate = 6000.0  # GB/s of 128-byte LINES the random-record gather sustained (diag.hip: gather32+cols, 2.16 TB/s of useful bytes = 160 B moved per 64 useful)
                c_ms, parts = 0.0, {}
                for t, k in zip(grp_tr, grp):
                    rate = line_rate if k["name"] == "k_hit_sub<gather>" else ach.get(pat.get(k["name"], "copy"), 5500.0)
                    ms = t * k["launches_per_step"] / (rate * 1e9) * 1e3
                    parts[k["name"]] = round(ms, 3)
                    c_ms += ms
                roof["ceiling"] = {"ms_per_step": round(c_ms, 3), "frac": round(grp_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "per_kernel_ms": parts,
                                   "what": "the sort group's counted traffic at the rates this GPU sustained for each kernel's access pattern (csrc/diag.hip, profiles/r04_diag_patterns.json): "
                                           "what this design can reach by SURVEY 8(d) bytes; the measured `frac` stands against THIS, 0.5 is out of its reach"}
            # the whole hit chain by the same accounting: SURVEY 8(d) sums the reference's passes to 584 B per stored hit
            chain_bytes = 584.0 * float(W.n_my)
            step_s = dt / args.steps
            roof["hit_chain"] = {"alg_bytes_per_step": chain_bytes, "achieved": round(chain_bytes / step_s / 1e9, 1), "unit": "GB/s",
                                 "frac": round(chain_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "584 B per stored hit (SURVEY 8(d), sum over the reference's passes) x hits of this rank / measured step time"}
    run.close()

    # ---- one input at a time: device + host latency of a single batch (no second context, no overlap with a neighbour): `ms_per_step` above is a
    # throughput figure with two inputs in flight
    latency = None
    if rank == 0 and world == 1 and want_leg("latency"):
        try:
            solo = Runner(W, overlap=False)
            solo.step(); solo.fence()
            lat = []
            for _ in range(3):
                t0 = time.perf_counter()
                solo.step(); solo.fence()
                lat.append((time.perf_counter() - t0) * 1e3)
            assert solo.output() == gfa
            solo.close()
            latency = {"ms": round(min(lat), 4), "ms_all": [round(x, 4) for x in lat], "what": "one input, records resident in HBM -> GFA text in host memory, nothing else in flight (best of 3)"}
        except Exception as e:
            log("latency leg failed:", e)

    # ---- the same job started one stage earlier: PAF TEXT resident in HBM -> device-side parse + dictionary -> ... -> GFA
    from_text = None
    buf, ln = vp(0), C.c_size_t(0)
    if want_text:
        try:
            d2 = L.sd_init()
            nh = C.c_size_t(0)

            def text_step():
                assert L.ma_hit_ingest_loaded(ctx.h, opt.min_span, opt.min_match, d2, C.byref(nh), 1, 0) == 0
                assert L.ma_pipeline_device_mem(ctx.h, C.byref(opt), d2, b"ug", 100, 0, C.byref(buf), C.byref(ln)) == 0
                n = ln.value
                L.free_buf(buf)
                return n
            n_txt = text_step()
            torch.cuda.synchronize()
            k = max(2, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(k):
                n_txt = text_step()
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t0
            assert n_txt == len(gfa) and nh.value == W.n_all, (n_txt, len(gfa), nh.value, W.n_all)
            from_text = {"value": total_lines * k / dtt, "unit": "overlaps/s", "ms_per_step": dtt / k * 1e3,
                         "input": "PAF text resident in HBM (%d bytes); each step parses it on the device, rebuilds the name dictionary on the host, runs the pipeline and writes the GFA" % W.size,
                         "file_to_hbm_s": W.t_load, "file_to_hbm_GBs": W.size / W.t_load / 1e9}
            L.mahip_paf_release(ctx.h)
            L.sd_destroy(d2)
        except Exception as e:
            log("from_text leg failed:", e)

    # ---- secondary legs (rank 0, one GPU)
    legs = {}
    if rank == 0 and world == 1 and not args.no_legs:
        def leg(name, reads, lines, seed, extra, steps, with_ref, prof_steps=0, gold_name=None):
            try:
                p = gen_paf(os.path.join(args.workdir, "leg_%s_r%d_n%d_s%d.paf" % (name, reads, lines, seed)), reads, lines, seed, extra)
                w = Workload(ma, L, ctx, p, opt, 1, 0)
                r = Runner(w)
                t = r.timed(1, steps)
                out = r.output()
                ti = ctx.tie_stats()
                res = {"value": w.n_lines * steps / t, "unit": "overlaps/s", "ms_per_step": t / steps * 1e3, "phases": r.phases(), "overlaps": w.n_lines, "reads": w.n_seq, "stored_hits": w.n_all,
                       "gfa_bytes": len(out), "gfa_md5": hashlib.md5(out).hexdigest(), "tie_groups": ti["arc_tie_groups"],
                       "tie_path": "arc walk%s" % (" + hit walk" if ti["hit_walk"] else "") if ti["arc_walk"] else "stable order (no arc ties)"}
                if prof_steps:  # HIP events around every timed scope of a few extra steps: where this input's time goes, and the reduce group's roofline
                    ctx.prof_enable(True)
                    ctx.prof_reset()
                    for _ in range(prof_steps):
                        r.step()
                        r.fence()  # (alone on the device: see the main workload's instrumented steps)
                    recs = ctx.prof_get()
                    ctx.prof_enable(False)
                    kt = kernel_table(recs, prof_steps)
                    L.mahip_asg_trans_inner.restype = C.c_uint64
                    L.mahip_asg_trans_inner.argtypes = [C.c_void_p]
                    n_inner = int(L.mahip_asg_trans_inner(ctx.h))
                    srt = next((k for k in kt if k["name"] == "k_arc_group_sort"), None) or next((k for k in kt if k["name"] == "k_asg_trans"), None)
                    n_arc = int(round(srt["alg_bytes_per_step"] / 48.0)) if srt and srt["name"] == "k_arc_group_sort" else None
                    res["arcs"] = n_arc
                    res["kernels"] = [{kk: k[kk] for kk in ("name", "launches_per_step", "avg_ms", "share", "alg_GBs", "design_GBs")} for k in kt[:16]]
                    res["reduce_group"] = reduce_group(kt, n_arc or 0, n_inner)
                    # counter traffic of this input's kernels, if a profile of THIS input is committed (the profile is of the default size: 100 M overlaps)
                    pmc_g, pmc_g_src = pmc_profile("gh") if (name == "graph_heavy" and lines == 100000000 and reads == 2000000) else (None, None)
                    if pmc_g and res["reduce_group"]:
                        tr = 0.0
                        for k in res["reduce_group"]["kernels"]:
                            t = pmc_traffic(pmc_g, k["name"])
                            k["counter_GBs"] = round(t / (k["avg_ms"] * 1e-3) / 1e9, 1) if t and k["avg_ms"] > 0 else None
                            tr += (t or 0) * k["launches_per_step"]
                        res["reduce_group"]["traffic"] = round(tr)
                        res["reduce_group"]["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py on this input; not measured in this run)" % pmc_g_src
                        res["reduce_group"]["frac_counter"] = round(tr / (res["reduce_group"]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                gold = recorded_reference(gold_name) if gold_name else None
                if gold and gold["pafgen"] == dict(reads=reads, lines=lines, seed=seed, extra=extra) and os.path.getsize(p) == gold["paf_bytes"] and head_tail_md5(p) == gold["paf_head_tail_md5"]:
                    # the unmodified reference's output on exactly this seeded text is on record (tests/golden/big.json, tests/golden/make_big.py): its minute of CPU is not spent again
                    res["gfa_identical"] = (hashlib.md5(out).hexdigest(), len(out)) == (gold["gfa_md5"], gold["gfa_bytes"])
                    res["reference"] = "recorded: %s (raw md5 of the reference's GFA on the same text; its wall there: %.1f s on %s)" % (gold_name, gold["reference_wall_s"], gold["host"])
                elif with_ref:
                    ref = run_reference(p, os.path.join(args.workdir, "leg_%s.ref.gfa" % name))
                    if ref:
                        res["gfa_identical"] = md5_pair(out)[0] == ref["md5"]
                        res["cpu_overlaps_per_s"] = w.n_lines / (ref["t_all"] - ref["t_parse"])
                r.close()
                w.close(L)
                legs[name] = res
            except Exception as e:
                log("leg %s failed:" % name, e)
        if cfg_name != "cfg2" and want_leg("cfg2"):
            leg("cfg2", 200000, 10000000, 1, [], 10, False)  # BASELINE configs[1]
        # coordinates on a 16-bp grid, 30 % dropout, 3 % false overlaps: thousands of equal (u,len) arc keys -- the default
        # mode finds them by census and reproduces the reference's (unstable-sort) order
        if want_leg("tie_rich"):
            leg("tie_rich", 250000, 5000000, 5, ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"], 3, not args.no_cpu)
        # graph-heavy ("always also run a graph-heavy fixed-length variant"): reads of ONE length, nothing is contained, every stored hit becomes an
        # arc -- the input on which arc sort, index, transitive reduction, symm and asg_arc_rm have work (at cfg4 containment leaves 1 M arcs of 200 M hits)
        if args.graph_heavy_lines > 0 and want_leg("graph_heavy"):
            leg("graph_heavy", max(args.graph_heavy_lines // 50, 100), args.graph_heavy_lines, 4, ["-L", "fixed"], 10, not args.no_cpu, prof_steps=2, gold_name="graph")  # (10 passes: the last pass's tail, 30 ms that nothing hides, is a tenth of the region)
            gold_g = recorded_reference("graph")
            if gold_g and "graph_heavy" in legs and args.graph_heavy_lines == gold_g["pafgen"]["lines"]:  # the same seeded text the digest file knows
                legs["graph_heavy"]["gfa_md5_matches_recorded_reference"] = legs["graph_heavy"].get("gfa_md5") == gold_g["gfa_md5"]
            if legs.get("graph_heavy", {}).get("reduce_group") and roof is not None:
                roof["reduce_group"] = dict(legs["graph_heavy"]["reduce_group"], input="legs.graph_heavy: pafgen -L fixed, %d overlaps, %d reads, %s arcs" % (
                    legs["graph_heavy"]["overlaps"], legs["graph_heavy"]["reads"], legs["graph_heavy"]["arcs"]))

    # ---- the reference on the same file: CPU baseline + the GFA every output above is compared with
    cpu, parity, e2e = None, None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            ref = run_reference(paf, os.path.join(args.workdir, "ref_%s.gfa" % cfg_name))
            if ref:
                mine = md5_pair(gfa)
                parity = {"gfa_identical": mine[0] == ref["md5"], "gfa_identical_sorted": mine[1] == ref["md5_sorted"], "gfa_md5": mine[0], "ref_md5": ref["md5"]}
                cpu = {"value": W.n_lines / (ref["t_all"] - ref["t_parse"]), "unit": "overlaps/s", "cores": 1, "kind": "reference",
                       "sample": "the SAME file (%d lines, %d reads); unmodified reference miniasm 0.3-r179 (gcc -O2), 1 thread pinned with taskset -c 0, 1 run; "
                                 "value = post-ingest part (its own stamps: total %.3f s - parse %.3f s), the part `value` measures; end to end incl. text parse: %.0f overlaps/s (%.1f s wall)" % (
                                     W.n_lines, W.n_seq, ref["t_all"], ref["t_parse"], W.n_lines / ref["t_all"], ref["wall"]),
                       "end_to_end_s": ref["wall"]}
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            log("cpu baseline failed:", e)
    if rank == 0 and world > 1:  # N ranks: no reference run here (65 s at cfg4), but the N = 1 run of the same work directory left its GFA behind
        try:
            ref_path = os.path.join(args.workdir, "ref_%s.gfa" % cfg_name)
            if os.path.exists(ref_path) and os.path.getmtime(ref_path) >= os.path.getmtime(paf):
                with open(ref_path, "rb") as f:
                    ref_md5 = md5_pair(f.read())
                mine = md5_pair(gfa)
                parity = {"gfa_identical": mine[0] == ref_md5[0], "gfa_identical_sorted": mine[1] == ref_md5[1], "gfa_md5": mine[0], "ref_md5": ref_md5[0],
                          "ref_from": "the reference's GFA of this file, kept by the N = 1 run in the work directory"}
        except Exception as e:
            log("parity against the cached reference GFA failed:", e)
    if early_e2e:
        best = min(early_e2e["walls"])
        e2e = {"value": W.n_lines / best, "unit": "overlaps/s", "wall_s": best, "wall_s_all": [round(x, 4) for x in early_e2e["walls"]],
               "what": "miniasm_amd/bin/miniasm <file> > out.gfa: process start to GFA on disk, warm page cache, best of 3 runs (3 s apart: see settle()), taken BEFORE this script holds device memory",
               "laps": early_e2e.get("laps"),
               "gfa_identical": (early_e2e["md5"] == parity["ref_md5"]) if parity else None,
               "gfa_md5_matches_recorded_reference": (early_e2e["md5"] == recorded_reference(cfg_name)["gfa_md5"]) if recorded_reference(cfg_name) else None,
               "vs_reference_wall": (cpu["end_to_end_s"] / best) if cpu else None}
    if early_cfg5:
        legs["cfg5"] = early_cfg5
    if early_real:
        legs["realistic"] = early_real
    if rank == 0:
        out = {
            "metric": "PAF overlaps processed/sec (hit-filter->trans-reduce->GFA)",
            "value": total_lines * args.steps / dt, "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s: synthetic %s PAF (pafgen seed %d), %d overlaps, %d reads, mean 8 kb, %.1f stored hits/read; inputs = unsorted 32-byte hit records resident in HBM%s; output = GFA text (%d bytes)" % (
                "BASELINE configs[3]" if cfg_name == "cfg4" else "BASELINE configs[1]" if cfg_name == "cfg2" else cfg_name,
                args.model, args.seed, W.n_lines, W.n_seq, W.n_all / max(W.n_seq, 1), " (sharded by query-read range)" if world > 1 else "", len(gfa)),
                "global_overlaps": W.n_lines, "per_gpu_hits": W.n_my,
                "pipelining": ("device tail (cleaners, unitigs) on a second context + host tail (GFA text) of pass k overlap the hit passes of pass k+1%s; all K outputs complete inside the timed region" % ("" if args.tail_one_stage else " (and each other: two worker threads)") if args.tail_ctx else
                               "host tail of pass k (GFA text) overlaps the device part of pass k+1; all K outputs complete inside the timed region") if overlap else "none (--no-overlap)",
                "parallelism": "read-range shards x%d, RCCL all-gather of sub/flags/arcs from C (host/sharded.c)" % world if world > 1 else "single GPU" + (", %d batches in flight" % args.inflight if args.inflight > 1 else "")},
            "gfa_identical": parity["gfa_identical"] if parity else None, "parity": parity,
            "tie_groups": tie["arc_tie_groups"] if tie else None,
            "tie_path": None if not tie else ("arc walk%s" % (" + hit walk" if tie["hit_walk"] else "") if tie["arc_walk"] else "unrepaired" if tie["unrepaired"] else "stable order (census: no arc ties => provably the reference's order)"),
            "phases": phases, "roofline": roof, "cpu_baseline": cpu, "latency": latency, "e2e": e2e, "from_text": from_text, "legs": legs, "kernels": kernels[:14],
            "setup": {"gen_s": t_gen, "file_to_hbm_s": W.t_load, "file_to_hbm_GBs": W.size / W.t_load / 1e9, "parse_dictionary_s": W.t_parse, "hbm_bytes_held": ctx.mem_bytes()},
        }
        if out["phases"] is None:
            out["phases"] = main_phases
        # the line's last object (a reader that only sees the tail of the line sees this): one number per leg
        rg = (roof or {}).get("reduce_group") or {}
        out["summary"] = {
            "ms_per_step": round(out["ms_per_step"], 4), "overlaps_per_s": round(out["value"]), "gfa_identical": out["gfa_identical"],
            "latency_ms": latency and latency["ms"], "from_text_ms_per_step": from_text and round(from_text["ms_per_step"], 3),
            "e2e_wall_s": e2e and round(e2e["wall_s"], 4), "e2e_vs_reference_wall": e2e and e2e["vs_reference_wall"] and round(e2e["vs_reference_wall"], 1),
            "roofline_frac_sort_group": roof and roof.get("frac"), "roofline_frac_counter_sort_group": roof and roof.get("frac_counter"),
            "roofline_frac_reduce_group": rg.get("frac"), "roofline_frac_counter_reduce_group": rg.get("frac_counter"),
            "cpu_reference_overlaps_per_s": cpu and round(cpu["value"]),
            "legs_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in legs.items() if isinstance(v, dict) and "ms_per_step" in v},
            "legs_wall_s": {k: round(v["wall_s"], 3) for k, v in legs.items() if isinstance(v, dict) and "wall_s" in v},
            "legs_identical": {k: v.get("gfa_identical", v.get("gfa_md5_matches_reference", v.get("gfa_md5_matches_recorded_reference"))) for k, v in legs.items() if isinstance(v, dict)},
        }
        print(json.dumps(out), flush=True)
    W.close(L)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
