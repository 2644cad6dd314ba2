"""Sharded multi-GPU mode of the hot path: one process per GPU, hits sharded by query-read range, RCCL exchanges
through torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The sequence below is the whole distributed algorithm:

    rank g owns reads [g*C, (g+1)*C), C = ceil(R / N), and every hit whose QUERY is one of them
    sort | sub1 | all-gather sub | [cut + flt + sub2 in one kernel] | all-gather sub2 | merge
    [cut + contained flags in one kernel] | max-all-reduce(r_cont), max-all-reduce(r_used) | squeeze map (identical on all ranks)
    sg candidate arcs | max-all-reduce(seq.del) | local rm + sort            (rank order = global (u,len) order)
    ALL-GATHER OF THE ARC BLOCKS  ->  every rank holds the whole sorted graph + CSR index
    transitive reduction of the rank's own vertices | all-gather of the del flags | rank 0: cleanup, symm, download

`backend` supplies the passes.  GpuBackend maps them 1:1 onto the C ABI (include/mahip.h); the CPU tests drive the
very same orchestration code over gloo with a test double built on the oracle (tests/test_dist_gloo.py).
All exchange buffers are flat uint8 torch tensors on the backend's device.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

BUF_SUB0, BUF_SUB1, BUF_RCONT, BUF_RUSED, BUF_SDEL = 0, 1, 2, 3, 4
_DEBUG = os.environ.get("MA_DEBUG_COMM") == "1"


class Comm:
    """torch.distributed collectives; degenerates to no-ops for a single rank"""

    def __init__(self, group=None):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.group = group
        # gloo (CPU tests, the one-GPU debug hook of bench.py) moves device buffers through the host; RCCL works on them in place
        self.via_host = self.on and dist.get_backend(group) == "gloo"

    def _stage(self, t):
        return t.cpu() if self.via_host and t.is_cuda else t

    def all_gather_bytes(self, local):
        """local: uint8 [n] -> uint8 [world*n] (rank-major)"""
        if self.world == 1:
            return local
        src = self._stage(local)
        out = torch.empty(self.world * src.numel(), dtype=torch.uint8, device=src.device)
        if _DEBUG:
            print("[comm %d] all_gather %d bytes on %s" % (self.rank, src.numel(), src.device), file=sys.stderr, flush=True)
        dist.all_gather_into_tensor(out, src, group=self.group)
        return out.to(local.device) if out.device != local.device else out

    def all_reduce_max_bytes(self, t):
        if self.world > 1:
            h = self._stage(t)
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            if h is not t:
                t.copy_(h)
        return t

    def all_gather_int(self, x, device):
        if self.world == 1:
            return [int(x)]
        t = torch.zeros(self.world, dtype=torch.int64, device="cpu" if self.via_host else device)
        t[self.rank] = int(x)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.tolist()]

    def sum_int(self, x, device):
        return sum(self.all_gather_int(x, device))

    def sum_ints(self, xs, device):
        """element-wise sum of a short list of integers over the ranks: one collective, one host sync"""
        if self.world == 1:
            return [int(x) for x in xs]
        t = torch.tensor([int(x) for x in xs], dtype=torch.int64, device="cpu" if self.via_host else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.tolist()]


class GpuBackend:
    """the passes on one MI355X through the C ABI; buffers are cuda uint8 tensors.

    Stream discipline: the mahip context is created ON a torch stream (GpuBackend.create) and run_sharded executes
    under torch.cuda.stream(that stream), so kernels, the device-to-device copies, torch tensor ops and the RCCL
    collectives (which torch orders against the current stream) are all ordered on ONE stream -- no host syncs are
    needed between a collective and the copy that consumes its result."""

    @classmethod
    def create(cls, device_index, n_seq):
        import miniasm_amd as ma
        dev = torch.device("cuda", device_index)
        stream = torch.cuda.Stream(device=dev)
        ctx = ma.Ctx(device_index, stream=C.c_void_p(stream.cuda_stream))
        be = cls(ctx, n_seq, dev)
        be.stream = stream
        return be

    stream = None

    def __init__(self, ctx, n_seq, device):
        import miniasm_amd as ma
        self.ma, self.L, self.ctx, self.h = ma, ma.lib(), ctx, ctx.h
        self.n_seq, self.device = n_seq, device
        L, vp, sz, u32, i32 = self.L, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
        L.mahip_copy_out.argtypes = [vp, i32, vp, sz, sz]
        L.mahip_copy_in.argtypes = [vp, i32, vp, sz, sz]
        L.mahip_hits_contained_flags.argtypes = [vp, C.POINTER(ma.MaOpt)]
        L.mahip_hits_contained_finish.argtypes = [vp, vp, C.POINTER(u32), C.POINTER(sz)]
        L.mahip_sg_flags.argtypes = [vp, C.POINTER(ma.MaOpt), i32, vp, vp]
        L.mahip_sg_finish.argtypes = [vp, C.POINTER(u32)]
        L.mahip_asg_export_rows.argtypes = [vp, vp]
        L.mahip_asg_import_rows.argtypes = [vp, vp, C.POINTER(u32), i32, sz]
        L.mahip_asg_del_trans_range.argtypes = [vp, i32, u32, u32, C.POINTER(u32)]
        L.mahip_asg_flags_out.argtypes = [vp, vp, sz, sz]
        L.mahip_asg_flags_in.argtypes = [vp, vp, sz, sz]
        L.mahip_asg_cleanup.argtypes = [vp, C.POINTER(u32)]
        L.mahip_hits_cutflt_sub.argtypes = [vp, i32, i32, i32, i32, i32, C.c_float, i32, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_float), C.POINTER(sz)]
        L.mahip_hits_cut_contained_flags.argtypes = [vp, i32, i32, C.POINTER(ma.MaOpt)]
        L.mahip_hits_cut_contained_finish.argtypes = [vp, C.POINTER(sz), C.POINTER(u32)]

    def _chk(self, rc, what):
        self.ma._chk(rc, what)

    def new_bytes(self, n):
        """exchange buffer of n bytes; buffers are kept and reused (every use is ordered on the backend's stream)"""
        n = max(int(n), 1)
        pool = self.__dict__.setdefault("_pool", {})
        k = pool.get("next", 0)
        pool["next"] = k + 1
        t = pool.get(k)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=torch.uint8, device=self.device)
            pool[k] = t
        return t[:n]

    def begin_pass(self):
        """a new pass may reuse the exchange buffers of the previous one, in the same order"""
        self.__dict__.setdefault("_pool", {})["next"] = 0

    def set_shard(self, q0, q1):
        self._chk(self.L.mahip_set_shard(self.h, q0, q1), "set_shard")

    def sort(self):
        self.ctx.sort()

    def sub(self, opt, slot, end_clip):
        return self.ctx.sub(opt.min_dp, opt.min_iden, end_clip, slot)

    def cut(self, opt, slot):
        return self.ctx.cut(slot, opt.min_span)

    def flt(self, opt, slot):
        return self.ctx.flt(slot, int(opt.max_hang * 1.5), int(opt.min_ovlp * .5))

    def merge(self):
        self.ctx.sub_merge()

    def cutflt_sub(self, opt):
        """first cut (against slot 0) + filter + second coverage pass (into slot 1) in one kernel; returns reads kept"""
        n_cut, n_flt, n_rem, cov = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_float(0)
        self._chk(self.L.mahip_hits_cutflt_sub(self.h, 0, opt.min_span, int(opt.max_hang * 1.5), int(opt.min_ovlp * .5), opt.min_dp, opt.min_iden,
                                               opt.min_span // 2, 1, C.byref(n_cut), C.byref(n_flt), C.byref(cov), C.byref(n_rem)), "cutflt_sub")
        return n_rem.value

    def cut_contained_flags(self, opt):
        """second cut (against slot 1) + containment flags (against the merged slot 0) in one kernel"""
        self._chk(self.L.mahip_hits_cut_contained_flags(self.h, 1, opt.min_span, C.byref(opt)), "cut_contained_flags")

    def cut_contained_finish(self):
        n, r = C.c_size_t(0), C.c_uint32(0)
        self._chk(self.L.mahip_hits_cut_contained_finish(self.h, C.byref(n), C.byref(r)), "cut_contained_finish")
        return r.value

    def hits_live(self):
        return int(self.L.mahip_hits_live(self.h))

    def copy_out(self, which, dst, first, count):
        self._chk(self.L.mahip_copy_out(self.h, which, dst.data_ptr(), first, count), "copy_out")

    def copy_in(self, which, src, first, count):
        self._chk(self.L.mahip_copy_in(self.h, which, src.data_ptr(), first, count), "copy_in")

    def contained_flags(self, opt):
        self._chk(self.L.mahip_hits_contained_flags(self.h, C.byref(opt)), "contained_flags")

    def contained_finish(self):
        n, r = C.c_size_t(0), C.c_uint32(0)
        self._chk(self.L.mahip_hits_contained_finish(self.h, None, C.byref(r), C.byref(n)), "contained_finish")
        return r.value, n.value

    def sg_flags(self, opt):
        self._chk(self.L.mahip_sg_flags(self.h, C.byref(opt), 1, None, None), "sg_flags")

    def sg_finish(self):
        n = C.c_uint32(0)
        self._chk(self.L.mahip_sg_finish(self.h, C.byref(n)), "sg_finish")
        return n.value

    def export_rows(self, dst):
        self._chk(self.L.mahip_asg_export_rows(self.h, dst.data_ptr()), "export_rows")

    def import_rows(self, src, counts, stride):
        arr = (C.c_uint32 * len(counts))(*counts)
        self._chk(self.L.mahip_asg_import_rows(self.h, src.data_ptr(), arr, len(counts), stride), "import_rows")

    def del_trans_range(self, opt, v0, v1):
        n = C.c_uint32(0)
        self._chk(self.L.mahip_asg_del_trans_range(self.h, opt.gap_fuzz, v0, v1, C.byref(n)), "del_trans_range")
        return n.value

    def flags_out(self, dst, first, count):
        self._chk(self.L.mahip_asg_flags_out(self.h, dst.data_ptr(), first, count), "flags_out")

    def flags_in(self, src, byte_off, first, count):
        self._chk(self.L.mahip_asg_flags_in(self.h, src.data_ptr() + byte_off, first, count), "flags_in")

    def cleanup(self):
        n = C.c_uint32(0)
        self._chk(self.L.mahip_asg_cleanup(self.h, C.byref(n)), "asg_cleanup")
        return n.value

    def symm(self):
        return self.ctx.symm()


def shard_range(n_seq, world, rank):
    c = (n_seq + world - 1) // world if world > 0 else n_seq
    q0 = min(n_seq, rank * c)
    return c, q0, min(n_seq, q0 + c)


def run_sharded(be, comm, opt, n_seq):
    """Drive one pass of the sharded pipeline up to the reduced graph.  Returns a dict of global counters; afterwards
    rank 0's backend holds the reduced graph (cleanup + symm done) ready for download."""
    stream = getattr(be, "stream", None)
    if stream is not None:
        with torch.cuda.stream(stream):
            return _run_sharded(be, comm, opt, n_seq)
    return _run_sharded(be, comm, opt, n_seq)


def _run_sharded(be, comm, opt, n_seq):
    N, g = comm.world, comm.rank
    Cn, q0, q1 = shard_range(n_seq, N, g)
    dev = be.device
    stats = {}
    if hasattr(be, "begin_pass"):
        be.begin_pass()
    be.set_shard(q0, q1)
    be.sort()

    def exchange_sub(slot):  # all-gather of the owned slices of a sub array (8 B per read)
        if N == 1:
            return
        loc = be.new_bytes(Cn * 8)
        be.copy_out(slot, loc, q0, q1 - q0)
        full = comm.all_gather_bytes(loc)
        be.copy_in(slot, full, 0, n_seq)

    def exchange_flags(*which):  # OR of 0/1 byte flags = max-all-reduce; several flag arrays share one collective
        if N == 1:
            return
        t = be.new_bytes(n_seq * len(which))
        for k, w in enumerate(which):
            be.copy_out(w, t[k * n_seq:(k + 1) * n_seq], 0, n_seq)
        comm.all_reduce_max_bytes(t)
        for k, w in enumerate(which):
            be.copy_in(w, t[k * n_seq:(k + 1) * n_seq], 0, n_seq)

    # counters are kept local and summed once at the end: every blocking exchange of a scalar costs a host sync
    loc_rem1 = be.sub(opt, 0, 0)
    exchange_sub(BUF_SUB0)
    loc_rem2 = be.cutflt_sub(opt)          # hit.c:162-216 + second ma_hit_sub: needs the complete first-pass intervals
    exchange_sub(BUF_SUB1)
    be.merge()                             # on the complete arrays, identical on every rank
    be.cut_contained_flags(opt)            # second cut + hit.c:225-245 flags for the local hits
    exchange_flags(BUF_RCONT, BUF_RUSED)
    stats["n_seq_new"] = be.cut_contained_finish()
    be.sg_flags(opt)
    exchange_flags(BUF_SDEL)
    n_loc = be.sg_finish()
    loc_hits = be.hits_live()
    if N > 1:  # the arc all-gather: blocks padded to the largest block
        counts = comm.all_gather_int(n_loc, dev)
        stride = max(max(counts), 1)
        rows = be.new_bytes(stride * 16)
        be.export_rows(rows)
        allrows = comm.all_gather_bytes(rows)
        be.import_rows(allrows, counts, stride)
    else:
        counts, stride = [n_loc], max(n_loc, 1)
    stats["n_arc"] = sum(counts)
    n_red = be.del_trans_range(opt, 2 * q0, 2 * q1)
    if N > 1:  # del flags of the own block -> everyone (only rank 0 needs them; kept symmetric)
        first = sum(counts[:g])
        fl = be.new_bytes(stride * 4)
        be.flags_out(fl, first, n_loc)
        allfl = comm.all_gather_bytes(fl)
        off = 0
        for r in range(N):
            if r != g and counts[r]:
                be.flags_in(allfl, r * stride * 4, off, counts[r])
            off += counts[r]
    loc_rem1, loc_rem2, loc_hits, n_red = comm.sum_ints([loc_rem1, loc_rem2, loc_hits, n_red], dev)
    stats["n_rem1"], stats["n_rem2"], stats["n_hits"], stats["n_red"] = loc_rem1, loc_rem2, loc_hits, n_red
    stats["n_multi"] = stats["n_asymm"] = 0
    if g == 0 and stats["n_red"]:
        be.cleanup()
        stats["n_multi"], stats["n_asymm"] = be.symm()
    return stats
