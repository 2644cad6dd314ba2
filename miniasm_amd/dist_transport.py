"""torch.distributed as the transport of the sharded mode's collectives (include/mahip.h: mahip_comm_init_ext): five callbacks on host buffers.

The product's data plane is RCCL called from C on the context's stream (csrc/comm.hip); this module exists so that the SAME C orchestration
(host/sharded.c, host/ingest_sharded.c) can run over any torch.distributed backend -- gloo on a CPU box: tests/test_dist_gloo.py drives it, world size
2 and 3, against the CPU build of the kernels.  There is no second copy of the exchange sequence in Python (rounds 1 - 3 kept one, miniasm_amd/sharded.py)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


class CommExt(C.Structure):  # include/mahip.h: mahip_comm_ext_t
    _fields_ = [("user", C.c_void_p),
                ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_reduce_max_u8", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_reduce_sum_u64", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_reduce_sum_u32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_to_all_v", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p))]


def _view(ptr, nbytes, dtype=np.uint8):
    """a numpy view of nbytes at ptr (no copy)"""
    if nbytes == 0:
        return np.zeros(0, dtype)
    return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype)


class Transport:
    """keeps the callback objects alive; .struct is what mahip_comm_init_ext takes"""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.calls = 0
        f = dict(CommExt._fields_)

        def guard(fn):
            def run(*a):
                try:
                    self.calls += 1
                    fn(*a)
                    return 0
                except Exception as e:  # never let an exception cross the C frame
                    print("[dist_transport] %s: %r" % (fn.__name__, e), flush=True)
                    return -1
            return run

        def all_gather(user, send, recv, nbytes):
            out = torch.from_numpy(_view(recv, nbytes * self.world))
            dist.all_gather_into_tensor(out, torch.from_numpy(_view(send, nbytes).copy()), group=self.group)

        def all_reduce_max_u8(user, buf, n):
            dist.all_reduce(torch.from_numpy(_view(buf, n)), op=dist.ReduceOp.MAX, group=self.group)

        def all_reduce_sum_u64(user, vals, n):
            t = torch.from_numpy(_view(vals, n * 8, np.int64))  # (two's complement sums of counters: as good as unsigned)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

        def all_reduce_sum_u32(user, buf, n):
            t = torch.from_numpy(_view(buf, n * 4, np.int32))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

        def all_to_all_v(user, send, recv, bytes_ptr):
            W, me = self.world, self.rank
            m = _view(bytes_ptr, W * W * 8, np.uint64).reshape(W, W)
            out_sizes, in_sizes = [int(m[me, j]) for j in range(W)], [int(m[i, me]) for i in range(W)]
            src = torch.from_numpy(_view(send, sum(out_sizes)).copy())
            dst = torch.from_numpy(_view(recv, sum(in_sizes)))
            outs, o = [], 0
            for k in out_sizes:
                outs.append(src[o:o + k]); o += k
            ins = [torch.empty(k, dtype=torch.uint8) for k in in_sizes]
            # gloo has no all_to_all for CPU tensors of unequal sizes on every build: W broadcasts-free rounds of paired send / recv
            reqs = []
            for step in range(W):
                to, frm = (me + step) % W, (me - step) % W
                if to == me:
                    ins[me].copy_(outs[me])
                    continue
                if out_sizes[to]:
                    reqs.append(dist.isend(outs[to].contiguous(), to, group=self.group))
                if in_sizes[frm]:
                    reqs.append(dist.irecv(ins[frm], frm, group=self.group))
            for r in reqs:
                r.wait()
            o = 0
            for i, k in enumerate(in_sizes):
                dst[o:o + k] = ins[i]; o += k

        self._cb = [f["all_gather"](guard(all_gather)), f["all_reduce_max_u8"](guard(all_reduce_max_u8)), f["all_reduce_sum_u64"](guard(all_reduce_sum_u64)),
                    f["all_reduce_sum_u32"](guard(all_reduce_sum_u32)), f["all_to_all_v"](guard(all_to_all_v))]
        self.struct = CommExt(None, *self._cb)

    def attach(self, L, ctx_handle):
        L.mahip_comm_init_ext.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CommExt)]
        rc = L.mahip_comm_init_ext(ctx_handle, self.rank, self.world, C.byref(self.struct))
        if rc != 0:
            raise RuntimeError("mahip_comm_init_ext failed")
