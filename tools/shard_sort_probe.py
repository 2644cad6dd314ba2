#!/usr/bin/env python3
"""probe: one process, one context -- the sharded head's first phases (sort on the own records, first coverage pass) for the shard of rank r of N, timed with
the phase marks.  Tells a slow sharded code path from an artefact of several processes sharing one GPU (tools/shard_projection.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: see tools/shard_projection.py)
torch.cuda.set_device(0)
import miniasm_amd as ma  # noqa: E402
import bench  # noqa: E402

L = ma.lib()
L.ma_set_log_path(b"/dev/null")
L.sys_init()
opt = ma.default_opt()
vp = C.c_void_p
paf = bench.gen_paf(os.path.join(os.environ.get("MA_BENCH_DIR", "/tmp/ma_bench"), "w_lognormal_r2000000_n100000000_s2.paf"), 2000000, 100000000, 2)
L.mahip_mark.argtypes = [vp, C.c_int]
L.mahip_marks_ms.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
L.mahip_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
L.mahip_set_full_input.argtypes = [vp, C.c_int]
L.mahip_hits_sort.argtypes = [vp]
L.mahip_hits_sub.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
for world, rank in ((1, 0), (2, 0), (2, 1), (8, 3)):
    ctx = ma.Ctx(0)
    W = bench.Workload(ma, L, ctx, paf, opt, world, rank)
    L.mahip_paf_release(ctx.h)
    per = (W.n_seq + world - 1) // world
    q0, q1 = min(rank * per, W.n_seq), min((rank + 1) * per, W.n_seq)
    for rep in range(3):
        ma._chk(L.mahip_hits_adopt(ctx.h, W.hits_dev.data_ptr(), W.n_my, W.n_seq), "adopt")
        L.mahip_set_hints(ctx.h, W.max_qs)
        if W.bounds is not None:  # a table of read ranges describes one upload (include/mahip.h): hand it over with every batch
            ma._chk(L.mahip_set_shard_bounds(ctx.h, W.bounds, len(W.bounds) - 1), "set_shard_bounds")
        L.mahip_set_full_input(ctx.h, 1 if world == 1 else 0)
        L.mahip_set_shard(ctx.h, q0 if world > 1 else 0, q1 if world > 1 else 0xffffffff)
        L.mahip_mark(ctx.h, 0)
        ma._chk(L.mahip_hits_sort(ctx.h), "sort")
        L.mahip_mark(ctx.h, 1)
        n_rem = C.c_size_t(0)
        ma._chk(L.mahip_hits_sub(ctx.h, opt.min_dp, opt.min_iden, 0, 0, C.byref(n_rem)), "sub")
        L.mahip_mark(ctx.h, 2)
        L.mahip_sync(ctx.h)
        ms = (C.c_float * 2)()
        L.mahip_marks_ms(ctx.h, 0, 2, ms)
    print("rank %d of %d: %d hits  sort %.3f ms  sub#1 %.3f ms" % (rank, world, W.n_my, ms[0], ms[1]), flush=True)
    W.close(L)
    ctx.close()
