#!/usr/bin/env python3
"""bench.py -- throughput of the miniasm hot path on MI355X.

Metric (BASELINE.json): PAF overlaps (input lines) processed per second through
    hit sort -> coverage/cut/filter x2 -> containment -> string graph -> transitive reduction + symm
    -> (host) tip/bubble/short-overlap cleaning -> unitigs -> GFA text,
with the parsed, unsorted 32-byte hit records already resident in HBM when the timed region starts (the text
ingest is host work outside the boundary; its rate and the PCIe-inclusive rate are reported in DESIGN.md).

Workload at N=1: BASELINE.json configs[1] -- synthetic 10M-overlap PAF, 200k reads, lognormal lengths with mean
8 kb, ~50 lines per read (miniasm_amd/bin/pafgen -r 200000 -n 10000000 -s 1).
N>1 (launched by torch.distributed.run, one rank per GPU): ONE data set of N x 10M overlaps / N x 200k reads, sharded by
query-read range; every rank runs the hit passes on its shard, sub / flag arrays and the arc blocks are exchanged over
RCCL (miniasm_amd/sharded.py), rank 0 finishes the graph and writes the GFA.  Weak scaling: per-GPU work is fixed;
value = global overlaps / max-over-ranks time.

One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
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


def cpu_baseline(args, workdir):
    """single-thread reference miniasm (oracle/_ref/miniasm_ref, unmodified) on a bounded sample of the same
    workload law, timed on this box's host cores; the post-ingest part (sort -> GFA) is what `value` measures."""
    import miniasm_amd as ma
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")
    reads, lines = args.reads // args.cpu_div, args.lines // args.cpu_div
    paf = gen_paf(os.path.join(workdir, "cpu_r%d_n%d_s%d.paf" % (reads, lines, args.seed + 1000)), reads, lines, args.seed + 1000, args.gen_extra)
    n_lines = 0
    with open(paf, "rb") as f:  # pafgen writes one overlap per line, newline-terminated
        while True:
            blk = f.read(64 << 20)
            if not blk:
                break
            n_lines += blk.count(b"\n")
    if os.path.exists(ref_bin):
        best = None
        for _ in range(args.cpu_runs):
            t0 = time.perf_counter()
            r = subprocess.run([ref_bin, paf], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
            m = re.search(r"\[M::ma_hit_read::([0-9.]+)\*", r.stderr)
            tot = re.search(r"Real time: ([0-9.]+) sec", r.stderr)
            if r.returncode != 0 or not m or not tot:
                return None
            t_parse, t_all = float(m.group(1)), float(tot.group(1))
            cur = (t_all - t_parse, t_all, wall)
            if best is None or cur[0] < best[0]:
                best = cur
        return {"value": n_lines / best[0], "unit": "overlaps/s", "cores": 1, "kind": "reference",
                "sample": "%d-line / %d-read sample of the same generator law; unmodified reference miniasm 0.3-r179 (gcc -O2), 1 thread, best of %d; "
                          "post-ingest part (its own stamps: total %.3f s - parse %.3f s); end-to-end incl. text parse %.0f overlaps/s" % (
                              n_lines, reads, args.cpu_runs, best[1], best[1] - best[0], n_lines / best[1])}
    # fallback: the C restatement (covers sort -> transitive reduction only)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stages as ST
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    t0 = time.perf_counter()
    ST.orc_stages(ing.hits, ing.n_seq, opt)
    dt = time.perf_counter() - t0
    ing.close()
    return {"value": n_lines / dt, "unit": "overlaps/s", "cores": 1, "kind": "port",
            "sample": "%d-line sample; oracle/ma_oracle.c (sort .. transitive reduction only, no cleaners/GFA), 1 thread" % n_lines}


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC run of this same workload (profiles/, made by
    `tools/gpu_round.sh pmc`: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950 -- the doubling reproduces the known 640 MB read of k_hit_keys exactly).  PMC counters
    cannot be read from inside this process, so the figure is only attached when the workload is the profiled one."""
    if (args.model, args.reads, args.lines) != ("lognormal", 200000, 10000000):
        return None
    path = os.path.join(ROOT, "profiles", "pmc_traffic_cfg2.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    names = {k.replace("void ", ""): v for k, v in d.items()}
    per_launch = lambda v: v["fetch_bytes_x2"] + v["write_bytes"]
    # a timed scope of the coverage passes = one launch of each size-class kernel: their bytes add up
    scope = {"k_hit_sub<cut+flt>": "k_hit_sub<true,", "k_hit_sub": "k_hit_sub<false,"}.get(kernel)
    if scope:
        parts = [v for k, v in names.items() if k.startswith(scope)]
        return round(sum(per_launch(v) for v in parts)) if parts else None
    base = kernel.split("<")[0]
    hits = [v for k, v in names.items() if k.split("<")[0] == base]
    if not hits:
        return None
    tot = sum(per_launch(v) * v["launches"] for v in hits) / max(sum(v["launches"] for v in hits), 1)
    return round(tot)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--lines", type=int, default=10000000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--model", default="lognormal", choices=["lognormal", "fixed", "uniform"])
    ap.add_argument("--workdir", default=os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-text", action="store_true", help="skip the text-resident leg (device-side parse)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-tie leg")
    ap.add_argument("--no-overlap", action="store_true", help="run each pass's host tail before the next pass's device part starts")
    ap.add_argument("--cpu-div", type=int, default=5, help="CPU baseline sample = workload / this")
    ap.add_argument("--cpu-runs", type=int, default=2)
    ap.add_argument("--prof-steps", type=int, default=3)
    args = ap.parse_args()
    args.gen_extra = [] if args.model == "lognormal" else ["-L", args.model]

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
        if one_gpu_debug:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
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

    # ---- setup (untimed): synthetic PAF text -> host ingest -> unsorted hit records into HBM
    # one global data set of world x (reads, lines); every rank parses it (same dictionary everywhere) and keeps the
    # hits whose query read falls into its range
    import numpy as np
    from miniasm_amd.sharded import Comm, GpuBackend, run_sharded, shard_range
    g_reads, g_lines = args.reads * world, args.lines * world
    t0 = time.perf_counter()
    paf = os.path.join(args.workdir, "w_%s_r%d_n%d_s%d.paf" % (args.model, g_reads, g_lines, args.seed))
    if rank == 0:
        gen_paf(paf, g_reads, g_lines, args.seed, args.gen_extra)
    if world > 1:
        dist.barrier()
    t_gen = time.perf_counter() - t0
    opt = ma.default_opt()
    t0 = time.perf_counter()
    ing = ma.Ingest(paf, opt)
    t_ingest = time.perf_counter() - t0
    n_lines = 0
    with open(paf, "rb") as f:  # pafgen writes one overlap per line, newline-terminated
        while True:
            blk = f.read(64 << 20)
            if not blk:
                break
            n_lines += blk.count(b"\n")
    n_seq = ing.n_seq
    _, q0, q1 = shard_range(n_seq, world, rank)
    if world > 1:
        q = (ing.hits["qns"] >> np.uint64(32)).astype(np.int64)
        my_hits = np.ascontiguousarray(ing.hits[(q >= q0) & (q < q1)])
        del q
    else:
        my_hits = ing.hits.copy()
    n_my, n_all = len(my_hits), ing.n
    ing.free_hits()
    hits_host = torch.from_numpy(my_hits.view("u1").reshape(-1))
    t0 = time.perf_counter()
    hits_dev = torch.empty(hits_host.numel(), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t_h2d = 1e-9
    if hits_host.numel():  # staged multi-threaded upload of the pageable records (include/mahip.h: mahip_memcpy_h2d)
        xc = ma.Ctx(local)  # the first queue of a process costs ~0.15 s: not part of the copy
        t0 = time.perf_counter()
        rc = ma.lib().mahip_memcpy_h2d(xc.h, C.c_void_p(hits_dev.data_ptr()), C.c_void_p(hits_host.data_ptr()), C.c_size_t(hits_host.numel()))
        t_h2d = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("mahip_memcpy_h2d: " + ma.lib().mahip_strerror().decode())
        xc.close()
    torch.cuda.synchronize()
    if rank == 0:
        log("workload: %d lines, %d stored hits (%d on this rank), %d reads; gen %.1fs ingest %.2fs (%.2f M lines/s) H2D %.3fs (%.1f GB/s)" % (
            n_lines, n_all, n_my, n_seq, t_gen, t_ingest, n_lines / t_ingest / 1e6, t_h2d, n_my * 32 / max(t_h2d, 1e-9) / 1e9))

    if world > 1:  # sharded mode: the context runs on a torch stream so RCCL collectives and kernels share one stream
        be = GpuBackend.create(local, n_seq)
        ctx = be.ctx
    else:
        ctx, be = ma.Ctx(local), None
    buf, ln = C.c_void_p(0), C.c_size_t(0)
    L.ma_pipeline_tail_mem.restype = C.c_int
    L.ma_pipeline_tail_mem.argtypes = [C.c_void_p, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    comm = Comm()

    max_qs = ing.max_qs

    # Passes are pipelined over the stream of batches: the device part of pass k+1 starts as soon as pass k's reduced graph
    # has been fetched (ma_pipeline_tail_fetch = the last use of the device for a batch), while pass k's host part (sequential
    # cleaners, unitigs, GFA text: ma_pipeline_tail_finish) runs on a worker thread.  All K outputs are complete before the
    # closing fence.  --no-overlap runs head and tail back to back.
    import queue
    import threading
    L.ma_pipeline_head.restype = C.c_int
    L.ma_pipeline_head.argtypes = [C.c_void_p, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_fetch.restype = C.c_void_p
    L.ma_pipeline_tail_fetch.argtypes = [C.c_void_p, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_finish_mem.restype = C.c_int
    L.ma_pipeline_tail_finish_mem.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    overlap = not args.no_overlap
    tail_out = {"n": 0, "rc": 0}

    def finish(job):
        b, l = C.c_void_p(0), C.c_size_t(0)
        rc = L.ma_pipeline_tail_finish_mem(job, C.byref(b), C.byref(l))
        tail_out["n"], tail_out["rc"] = l.value, tail_out["rc"] or rc
        L.free_buf(b)

    tailq = queue.Queue(maxsize=1)  # one batch may wait while another is being finished

    def tail_worker():
        while True:
            job = tailq.get()
            try:
                if job is not None:
                    finish(job)
            except Exception as e:  # never leave the fence waiting on a dead worker
                tail_out["rc"] = tail_out["rc"] or -1
                log("host tail failed:", e)
            finally:
                tailq.task_done()
            if job is None:
                return

    worker = None
    if overlap and rank == 0:
        worker = threading.Thread(target=tail_worker, daemon=True)
        worker.start()

    def step():
        ma._chk(L.mahip_hits_adopt(ctx.h, hits_dev.data_ptr(), n_my, n_seq), "adopt")
        L.mahip_set_hints(ctx.h, max_qs)
        st = (C.c_uint32 * 4)(0, 0, 0, 0)
        if world == 1:  # single GPU: the C pipeline's device half
            assert L.ma_pipeline_head(ctx.h, C.byref(opt), ing.d, b"ug", 100, 0, C.byref(st)) == 0
        else:  # sharded: device passes + RCCL exchanges on every rank, graph cleaning + GFA on rank 0
            stats = run_sharded(be, comm, opt, n_seq)
            if rank != 0:
                return 0
            st = (C.c_uint32 * 4)(1, 1, stats["n_red"], 1)
        job = L.ma_pipeline_tail_fetch(ctx.h, C.byref(opt), ing.d, b"ug", 100, C.byref(st))
        assert job
        if worker:
            tailq.put(job)
        else:
            finish(job)
        return tail_out["n"]

    def fence():
        if worker:
            tailq.join()  # every host tail handed over so far is complete
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        assert tail_out["rc"] == 0

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    gfa_len = tail_out["n"]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu_debug else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    total_lines = float(n_lines)  # the global data set (world x per-GPU lines)

    # ---- per-kernel timing with HIP events on the launch stream (separate, instrumented steps)
    roof, kernels = None, []
    if rank == 0:
        ctx.prof_enable(True)
        ctx.prof_reset()
    for _ in range(args.prof_steps):  # a step is collective in the sharded mode: EVERY rank runs it, rank 0 is the one instrumented
        step()
    fence()
    if rank == 0:
        recs = ctx.prof_get()
        ctx.prof_enable(False)
        tot_ms = sum(r["total_ms"] for r in recs) or 1.0
        for r in sorted(recs, key=lambda r: -r["total_ms"]):
            per = r["total_ms"] / max(r["launches"], 1)
            kernels.append({"name": r["name"], "launches_per_step": r["launches"] / args.prof_steps, "avg_ms": round(per, 5),
                            "share": round(r["total_ms"] / tot_ms, 4),
                            "alg_GBs": round(r["alg_bytes"] / max(r["launches"], 1) / (per * 1e-3) / 1e9, 1) if per > 0 and r["alg_bytes"] > 0 else None})
        dom = next((k for k in kernels if k["alg_GBs"]), None)
        if dom:
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["alg_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom["alg_GBs"] / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom["name"], args),
                    "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"]}

    # ---- the same job started one stage earlier: PAF TEXT resident in HBM -> device-side parse + dictionary -> ... -> GFA
    from_text = None
    if rank == 0 and world == 1 and not args.no_text:
        try:
            L.ma_paf_load_file.argtypes = [C.c_void_p, C.c_char_p]
            L.ma_hit_ingest_loaded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(ma.Sdict), C.POINTER(C.c_size_t), C.c_int, C.c_int]
            t0 = time.perf_counter()
            assert L.ma_paf_load_file(ctx.h, paf.encode()) == 0
            t_load = time.perf_counter() - t0
            d2 = L.sd_init()
            nh = C.c_size_t(0)

            def text_step():
                assert L.ma_hit_ingest_loaded(ctx.h, opt.min_span, opt.min_match, d2, C.byref(nh), 1, 0) == 0
                assert L.ma_pipeline_device_mem(ctx.h, C.byref(opt), d2, b"ug", 100, 0, C.byref(buf), C.byref(ln)) == 0
                n = ln.value
                L.free_buf(buf)
                return n
            for _ in range(args.warmup):
                n_txt = text_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                n_txt = text_step()
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t0
            assert n_txt == gfa_len and nh.value == n_all, (n_txt, gfa_len, nh.value, n_all)
            from_text = {"value": total_lines * args.steps / dtt, "unit": "overlaps/s", "ms_per_step": dtt / args.steps * 1e3,
                         "input": "PAF text resident in HBM (%d bytes); each step parses it on the device, rebuilds the name dictionary on the host, runs the pipeline and writes the GFA" % os.path.getsize(paf),
                         "file_to_hbm_s": t_load, "file_to_hbm_GBs": os.path.getsize(paf) / t_load / 1e9}
            L.mahip_paf_release(ctx.h)
            L.sd_destroy(d2)
        except Exception as e:
            log("from_text leg failed:", e)

    # ---- the same job with the reference's order of equal sort keys (guaranteed byte-identical output on ANY input: the tie
    # order is a sequential function of the whole input, computed on the host from the keys)
    exact = None
    if rank == 0 and world == 1 and not args.no_exact:
        try:
            L.mahip_set_exact_ties(ctx.h, 1)

            def exact_step():
                ma._chk(L.mahip_hits_adopt(ctx.h, hits_dev.data_ptr(), n_my, n_seq), "adopt")
                L.mahip_set_hints(ctx.h, max_qs)
                assert L.ma_pipeline_device_mem(ctx.h, C.byref(opt), ing.d, b"ug", 100, 0, C.byref(buf), C.byref(ln)) == 0
                n = ln.value
                L.free_buf(buf)
                return n
            n_ex = exact_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                n_ex = exact_step()
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / 2
            exact = {"value": total_lines / dte, "unit": "overlaps/s", "ms_per_step": dte * 1e3, "gfa_bytes": n_ex,
                     "note": "MA_EXACT_TIES=1: host emulation of the reference's unstable radix sort for the hit and arc orders (8 B/hit down, 4 B/hit up)"}
        except Exception as e:
            log("exact-tie leg failed:", e)
        finally:
            L.mahip_set_exact_ties(ctx.h, 0)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            cpu = cpu_baseline(args, args.workdir)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            log("cpu baseline failed:", e)

    if rank == 0:
        out = {
            "metric": "PAF overlaps processed/sec (hit-filter->trans-reduce->GFA)",
            "value": total_lines * args.steps / dt, "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "synthetic %s PAF: %d overlaps, %d reads, mean 8 kb, %.1f stored hits/read, seed %d; inputs = unsorted 32-byte hit records resident in HBM%s; output = GFA text (%d bytes)" % (
                args.model, n_lines, n_seq, n_all / max(n_seq, 1), args.seed, " (sharded by query-read range)" if world > 1 else "", gfa_len),
                "per_gpu_overlaps": n_lines // world,
                "pipelining": "host tail of pass k (cleaners, unitigs, GFA text) overlaps the device part of pass k+1; all K outputs complete inside the timed region" if overlap else "none (--no-overlap)",
                "parallelism": "read-range shards x%d, RCCL all-gather of sub/flags/arcs" % world if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu, "from_text": from_text, "exact_ties": exact, "kernels": kernels[:12],
            "setup": {"ingest_lines_per_s": n_lines / t_ingest, "h2d_GBs": n_my * 32 / max(t_h2d, 1e-9) / 1e9, "hbm_bytes_held": ctx.mem_bytes()},
        }
        print(json.dumps(out), flush=True)
    ing.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
