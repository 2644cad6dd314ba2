"""One rank of a bench.py-shaped N-rank run, without torch: this rank holds ONLY the records of its read range (full_input = 0), the
collectives go through the shared-memory double, rank 0 finishes every batch -- directly or on a second context (mahip_tail_handoff) -- and
compares the GFA of every step with the single-context run.  Started by tests/test_gpu_sharded.py, once per rank."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("MA_WORKER_EMU") == "1":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu_plugin  # noqa: F401
import miniasm_amd as ma  # noqa: E402


def main():
    paf, rank, world, name, tail_ctx, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1", sys.argv[6]
    L = ma.lib()
    vp = C.c_void_p
    L.ma_set_log_path(b"/dev/null")
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    n_seq = ing.n_seq
    per = (n_seq + world - 1) // world
    q0, q1 = min(rank * per, n_seq), min((rank + 1) * per, n_seq)
    bounds = None
    if os.environ.get("MA_WORKER_BALANCE") == "1":  # read ranges with equally many hits (mahip_hits_balance, as bench.py and the command line use them)
        L.mahip_hits_balance.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        tmp = ma.Ctx(0)
        tmp.hits_upload(ing.hits, n_seq)
        bounds = (C.c_uint32 * (world + 1))()
        ma._chk(L.mahip_hits_balance(tmp.h, world, bounds), "balance")
        tmp.close()
        q0, q1 = bounds[rank], bounds[rank + 1]
        assert bounds[0] == 0 and bounds[world] == n_seq and all(bounds[r] <= bounds[r + 1] for r in range(world))
    qid = (ing.hits["qns"] >> np.uint64(32)).astype(np.int64)
    sel = (qid >= q0) & (qid < q1)
    mine = np.ascontiguousarray(ing.hits[sel])  # this rank's records, input order kept
    pos = np.ascontiguousarray(np.nonzero(sel)[0].astype(np.uint32)) if os.environ.get("MA_WORKER_POS") == "1" else None  # ... and where they stood

    ShardStats = ma.ShardStats
    L.ma_pipeline_head_sharded.restype = C.c_int
    L.ma_pipeline_head_sharded.argtypes = [vp, C.POINTER(ma.MaOpt), C.c_uint32, C.c_int, C.POINTER(ShardStats)]
    L.mahip_comm_init_shm.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.mahip_comm_destroy.argtypes = [vp]
    L.mahip_tail_handoff.argtypes = [vp, vp]
    L.ma_pipeline_tail_mem.restype = C.c_int
    L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
    ctx = ma.Ctx(0)
    ctx2 = ma.Ctx(0) if (tail_ctx and rank == 0) else None
    ma._chk(L.mahip_comm_init_shm(ctx.h, name.encode(), rank, world), "comm_init_shm")
    L.mahip_set_shard_bounds.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int]
    outs = []
    for step in range(3):
        ctx.hits_upload(mine, n_seq)
        if bounds is not None:  # a table of read ranges describes one upload: every upload forgets it
            ma._chk(L.mahip_set_shard_bounds(ctx.h, bounds, world), "set_shard_bounds")
        if pos is not None:  # positions describe one upload: with them the ranks can restore the reference's order of tied hits (mahip_hits_set_positions)
            L.mahip_hits_set_positions.argtypes = [vp, C.c_void_p, C.c_int, C.c_uint64]
            ma._chk(L.mahip_hits_set_positions(ctx.h, pos.ctypes.data, 0, len(ing.hits)), "set_positions")
        stats = ShardStats()
        assert L.ma_pipeline_head_sharded(ctx.h, C.byref(opt), n_seq, 0, C.byref(stats)) == 0
        if pos is not None:
            assert stats.tie_groups == 0 or stats.tie_repaired == 1, "tie groups left unrepaired although every rank knows its positions"
        if rank != 0:
            continue
        st = (C.c_uint32 * 4)(1, 1, stats.n_red, 1)
        tail_on = ctx
        if ctx2 is not None:
            ma._chk(L.mahip_tail_handoff(ctx.h, ctx2.h), "tail_handoff")
            tail_on = ctx2
        buf, ln = vp(0), C.c_size_t(0)
        assert L.ma_pipeline_tail_mem(tail_on.h, C.byref(opt), ing.d, b"ug", 100, C.byref(st), C.byref(buf), C.byref(ln)) == 0
        outs.append(C.string_at(buf, ln.value))
        L.free_buf(buf)
    L.mahip_comm_destroy(ctx.h)
    if rank == 0:
        one = ma.Ctx(0)
        one.hits_upload(ing.hits, n_seq)
        want = ma.run_resident(one, opt, ing, "ug")
        ok = all(o == want for o in outs) and len(outs) == 3
        with open(out_path, "wb") as f:
            f.write(b"OK\n" if ok else b"DIFFERENT\n")
            f.write(want)
        one.close()
    ing.close()
    ctx.close()
    if ctx2 is not None:
        ctx2.close()


if __name__ == "__main__":
    main()
