#!/usr/bin/env python3
"""One-GPU PROJECTION of the sharded step (not a measurement of scaling): N processes share the one GPU of the box through the shared-memory
double of the collectives and TAKE TURNS on it (MA_SHM_SERIAL=1: a rank holds the device while it computes and gives it up for the length of a
collective), so the device time of each phase of host/sharded.c (HIP events between the phases, mahip_mark) is what the rank would see on a GPU
of its own.  Per N:  predicted step = sum over the compute phases of the slowest rank's time  +  sum over the exchanges of (latency + bytes a rank
receives / link bandwidth)  [+ rank 0's tail when it is not hidden behind the next step's head].
usage: tools/shard_projection.py [--reads R --lines N --seed S] [--ranks 1,2,4,8] [--steps 3] [--out gpurun_out/shard_projection.json]"""
import argparse
import ctypes as C
import fcntl
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LINK_GBS = 153.0   # one xGMI link, one direction (7 per GPU); a ring all-gather is bound by one link in, one link out
LAT_US = 25.0      # per collective


def worker(a):
    import torch  # first: its HIP runtime has to be the one the process initialises (bench.py's order), or torch sees no device afterwards
    if not torch.cuda.is_available():
        sys.exit("shard_projection.py needs a GPU")
    torch.cuda.set_device(0)
    import miniasm_amd as ma
    import bench
    L = ma.lib()
    L.ma_set_log_path(b"/dev/null")
    L.sys_init()
    opt = ma.default_opt()
    vp = C.c_void_p
    ctx = ma.Ctx(0)
    with open(a.paf + ".lock", "w") as lk:  # one rank at a time through load + parse (each holds the whole text for a moment)
        fcntl.flock(lk, fcntl.LOCK_EX)
        W = bench.Workload(ma, L, ctx, a.paf, opt, a.world, a.rank)
        L.mahip_paf_release(ctx.h)
        L.mahip_mem_trim.argtypes = [C.c_void_p, C.c_void_p]
        L.mahip_mem_trim(ctx.h, None)  # the ranks share ONE GPU here: the text and the parser's columns (20 GB at cfg4) must not sit idle in N pools
        L.mahip_sync(ctx.h)
        fcntl.flock(lk, fcntl.LOCK_UN)
    L.ma_pipeline_head_sharded.restype = C.c_int
    L.ma_pipeline_head_sharded.argtypes = [vp, C.POINTER(ma.MaOpt), C.c_uint32, C.c_int, C.POINTER(ma.ShardStats)]
    L.mahip_comm_init_shm.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.ma_pipeline_tail_mem.restype = C.c_int
    L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ma_shard_phases.argtypes = [C.c_int]
    L.mahip_comm_destroy.argtypes = [vp]
    if a.world > 1:
        ma._chk(L.mahip_comm_init_shm(ctx.h, a.name.encode(), a.rank, a.world), "comm_init_shm")
    L.ma_shard_phases(1)
    rows, tails, md5 = [], [], None
    for step in range(a.steps):
        ma._chk(L.mahip_hits_adopt(ctx.h, W.hits_dev.data_ptr(), W.n_my, W.n_seq), "adopt")
        L.mahip_set_hints(ctx.h, W.max_qs)
        if W.bounds is not None:  # a table of read ranges describes one upload (include/mahip.h): hand it over with every batch
            ma._chk(L.mahip_set_shard_bounds(ctx.h, W.bounds, len(W.bounds) - 1), "set_shard_bounds")
        st = ma.ShardStats()
        assert L.ma_pipeline_head_sharded(ctx.h, C.byref(opt), W.n_seq, 0, C.byref(st)) == 0
        rows.append({"phase_ms": [float(x) for x in st.phase_ms], "xchg_bytes": [int(x) for x in st.xchg_bytes]})
        if a.rank == 0:
            s4 = (C.c_uint32 * 4)(1, 1, st.n_red, 1)
            buf, ln = vp(0), C.c_size_t(0)
            t0 = time.perf_counter()
            assert L.ma_pipeline_tail_mem(ctx.h, C.byref(opt), W.d, b"ug", 100, C.byref(s4), C.byref(buf), C.byref(ln)) == 0
            tails.append((time.perf_counter() - t0) * 1e3)
            import hashlib
            md5 = hashlib.md5(C.string_at(buf, ln.value)).hexdigest()
            L.free_buf(buf)
    json.dump({"rank": a.rank, "world": a.world, "n_my": W.n_my, "n_all": W.n_all, "n_seq": W.n_seq, "n_lines": W.n_lines, "steps": rows, "tail_ms": tails, "gfa_md5": md5}, open(a.out, "w"))
    if a.world > 1:
        L.mahip_comm_destroy(ctx.h)
    W.close(L)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2000000)
    ap.add_argument("--lines", type=int, default=100000000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--workdir", default=os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shard_projection.json"))
    ap.add_argument("--per-n-timeout", type=float, default=240.0, help="seconds one N may take before its ranks are killed")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--paf"); ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int); ap.add_argument("--name")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    import miniasm_amd as ma
    import bench
    os.makedirs(a.workdir, exist_ok=True)
    paf = bench.gen_paf(os.path.join(a.workdir, "w_lognormal_r%d_n%d_s%d.paf" % (a.reads, a.lines, a.seed)), a.reads, a.lines, a.seed)
    names = ma.SHARD_PHASE_NAMES
    result = {"what": __doc__.split("usage:")[0].strip(), "workload": "pafgen -r %d -n %d -s %d" % (a.reads, a.lines, a.seed),
              "model": {"link_GBs": LINK_GBS, "latency_us_per_collective": LAT_US, "all_gather_time": "latency + (N-1)/N x bytes received / link_GBs"}, "runs": []}
    base = None
    for n in [int(x) for x in a.ranks.split(",")]:
        env = dict(os.environ, MA_SHM_SERIAL="1")
        procs, outs = [], []
        for r in range(n):
            o = os.path.join(a.workdir, "proj_%d_%d.json" % (n, r))
            outs.append(o)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", "--paf", paf, "--rank", str(r), "--world", str(n),
                                           "--name", "ma_proj_%d_%d" % (os.getpid(), n), "--steps", str(a.steps), "--out", o], env=env))
        rc = []
        t_end = time.time() + a.per_n_timeout
        for p in procs:  # a run that does not come back (round 4, visit D: N = 8 sat until the visit's time limit) is ended here, not by the caller's budget
            try:
                rc.append(p.wait(timeout=max(1.0, t_end - time.time())))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                rc.append(-9)
        if any(rc):
            print("N=%d: a rank failed (%r)" % (n, rc), file=sys.stderr)
            continue
        ranks = [json.load(open(o)) for o in outs]
        # per rank and phase: the best of the steps behind the first (the ranks take turns on one GPU; a turn that collides with another process's work on the box
        # shows as one slow phase on one rank of one step -- round 4, visit C: "sort" 1.85 ms on one rank of eight, 0.82 on the others); then the slowest rank
        last = [rk["steps"][-1] for rk in ranks]
        best = [[min(st["phase_ms"][i] for st in rk["steps"][1:] or rk["steps"]) for i in range(len(names))] for rk in ranks]
        pmax = [max(b[i] for b in best) for i in range(len(names))]
        pmin = [min(b[i] for b in best) for i in range(len(names))]
        xb = [max(l["xchg_bytes"][i] for l in last) for i in range(len(names))]
        compute = sum(pmax[i] for i, nm in enumerate(names) if not nm.startswith("x:"))
        n_coll = sum(1 for i, nm in enumerate(names) if nm.startswith("x:")) if n > 1 else 0
        xchg = sum(LAT_US * 1e-3 + (n - 1) / n * xb[i] / (LINK_GBS * 1e9) * 1e3 for i, nm in enumerate(names) if nm.startswith("x:")) if n > 1 else 0.0
        tail = ranks[0]["tail_ms"][-1]
        run = {"n_ranks": n, "hits_per_rank": [rk["n_my"] for rk in ranks], "gfa_md5": ranks[0]["gfa_md5"],
               "phase_ms_max_min": {nm: [round(pmax[i], 4), round(pmin[i], 4)] for i, nm in enumerate(names)},
               "exchange_bytes_received_per_rank": {nm: xb[i] for i, nm in enumerate(names) if xb[i]},
               "compute_ms_slowest_rank_per_phase": round(compute, 4), "collectives": n_coll, "exchange_ms_model": round(xchg, 4), "rank0_tail_ms_alone": round(tail, 4),
               "predicted_step_ms_tail_hidden": round(compute + xchg, 4), "predicted_step_ms_tail_serial": round(compute + xchg + tail, 4)}
        if base is None:
            base = run
        run["predicted_speedup_vs_1_tail_hidden"] = round(base["predicted_step_ms_tail_hidden"] / run["predicted_step_ms_tail_hidden"], 3)
        run["predicted_speedup_vs_1_tail_serial"] = round(base["predicted_step_ms_tail_serial"] / run["predicted_step_ms_tail_serial"], 3)
        run["same_gfa_as_1_rank"] = run["gfa_md5"] == base["gfa_md5"]
        result["runs"].append(run)
        print("N=%d  compute %.3f ms  exchanges(model) %.3f ms  rank-0 tail %.3f ms  -> step %.3f ms (tail hidden) / %.3f (serial); x%.2f / x%.2f vs N=1; same GFA: %s" % (
            n, compute, xchg, tail, run["predicted_step_ms_tail_hidden"], run["predicted_step_ms_tail_serial"], run["predicted_speedup_vs_1_tail_hidden"],
            run["predicted_speedup_vs_1_tail_serial"], run["same_gfa_as_1_rank"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(result, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
