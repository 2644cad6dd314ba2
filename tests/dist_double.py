"""CPU stand-in for the GPU backend of miniasm_amd/sharded.py, built on the C oracle.  It lets the CPU tests drive
the REAL orchestration code (shard ranges, exchange points, padding, offsets) over gloo with world_size > 1.
Semantics mirror the device: read ids keep their original numbering (the squeeze map is only applied when the
result is exported), every rank holds only the hits whose query lies in its read range."""
import ctypes as C

import numpy as np
import torch

import miniasm_amd as ma
import refapi as R
from miniasm_amd.sharded import BUF_RCONT, BUF_RUSED, BUF_SDEL, BUF_SUB0, BUF_SUB1

HIT_DT, SUB_DT, ARC_DT = ma.HIT_DT, ma.SUB_DT, ma.ARC_DT


def _p(a):
    return a.ctypes.data


class OracleBackend:
    device = torch.device("cpu")

    def __init__(self, hits, n_seq):
        self.O = R.orc()
        O, vp, sz, u32, i32 = self.O, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
        O.orc_contained_flags.argtypes = [C.POINTER(ma.MaOpt), vp, sz, vp, vp, vp]
        O.orc_sg_candidates.restype = sz
        O.orc_sg_candidates.argtypes = [C.POINTER(ma.MaOpt), u32, vp, vp, vp, sz, vp, vp, vp, vp]
        O.orc_sg_finish.restype = sz
        O.orc_sg_finish.argtypes = [sz, vp, vp]
        O.orc_arc_del_trans_range.restype = u32
        O.orc_arc_del_trans_range.argtypes = [u32, sz, vp, vp, vp, i32, u32, u32, C.POINTER(C.c_uint64)]
        self.all_hits, self.R = np.ascontiguousarray(hits, dtype=HIT_DT), n_seq
        self.subs = [np.zeros(n_seq, SUB_DT), np.zeros(n_seq, SUB_DT)]
        self.flag = {BUF_RCONT: np.zeros(n_seq, np.uint8), BUF_RUSED: np.zeros(n_seq, np.uint8), BUF_SDEL: np.zeros(n_seq, np.uint8)}
        self.r_del = np.zeros(n_seq, np.uint8)

    def new_bytes(self, n):
        return torch.zeros(max(int(n), 1), dtype=torch.uint8)

    def _arr(self, which):
        return self.subs[which].view(np.uint8) if which in (BUF_SUB0, BUF_SUB1) else self.flag[which]

    def _es(self, which):
        return 8 if which in (BUF_SUB0, BUF_SUB1) else 1

    def copy_out(self, which, dst, first, count):
        es = self._es(which)
        dst.numpy()[:count * es] = self._arr(which)[first * es:(first + count) * es]

    def copy_in(self, which, src, first, count):
        es = self._es(which)
        self._arr(which)[first * es:(first + count) * es] = src.numpy()[:count * es]

    def set_shard(self, q0, q1):
        q = (self.all_hits["qns"] >> 32).astype(np.int64)
        self.hits = self.all_hits[(q >= q0) & (q < q1)].copy()

    def sort(self):
        self.O.orc_hit_sort(len(self.hits), _p(self.hits))

    def sub_(self, slot):
        return self.subs[slot]

    def sub(self, opt, slot, end_clip):
        self.subs[slot][:] = 0
        return self.O.orc_hit_sub(opt.min_dp, opt.min_iden, end_clip, len(self.hits), _p(self.hits), self.R, _p(self.subs[slot]))

    def cut(self, opt, slot):
        n = self.O.orc_hit_cut(_p(self.subs[slot]), opt.min_span, len(self.hits), _p(self.hits))
        self.hits = self.hits[:n].copy()
        return n

    def flt(self, opt, slot):
        cov = C.c_float(0)
        n = self.O.orc_hit_flt(_p(self.subs[slot]), int(opt.max_hang * 1.5), int(opt.min_ovlp * .5), len(self.hits), _p(self.hits), C.byref(cov))
        self.hits = self.hits[:n].copy()
        return n, cov.value

    def merge(self):
        self.O.orc_sub_merge(self.R, _p(self.subs[0]), _p(self.subs[1]))

    def contained_flags(self, opt):
        self.flag[BUF_RCONT][:] = 0
        self.flag[BUF_RUSED][:] = 0
        self.O.orc_contained_flags(C.byref(opt), _p(self.subs[0]), len(self.hits), _p(self.hits), _p(self.flag[BUF_RCONT]), _p(self.flag[BUF_RUSED]))

    def contained_finish(self):
        sdel = (self.subs[0]["sdel"] >> 31).astype(np.uint8)
        self.r_del = (sdel | self.flag[BUF_RCONT] | (1 - self.flag[BUF_RUSED])).astype(np.uint8)
        keep = 1 - self.r_del.astype(np.int64)
        self.map = np.where(self.r_del == 0, np.cumsum(keep) - 1, -1).astype(np.int32)
        q, t = (self.hits["qns"] >> 32).astype(np.int64), self.hits["tn"].astype(np.int64)
        self.hits = self.hits[(self.r_del[q] == 0) & (self.r_del[t] == 0)].copy()
        return int(keep.sum()), len(self.hits)

    # the fused entries of the product (one kernel each there) as compositions of the oracle's passes
    def cutflt_sub(self, opt):
        self.cut(opt, 0)
        self.flt(opt, 0)
        return self.sub(opt, 1, opt.min_span // 2)

    def cut_contained_flags(self, opt):  # the orchestration has merged the intervals already: cut against slot 1, classify against slot 0
        self.cut(opt, 1)
        self.contained_flags(opt)

    def cut_contained_finish(self):
        return self.contained_finish()[0]

    def hits_live(self):
        return len(self.hits)

    def sg_flags(self, opt):
        n = len(self.hits)
        self.cand = np.zeros(max(n, 1), ARC_DT)
        self.slen = np.zeros(max(self.R, 1), "<u4")
        self.flag[BUF_SDEL][:] = 0
        self.n_cand = self.O.orc_sg_candidates(C.byref(opt), self.R, _p(self.subs[0]), None, _p(self.r_del), n, _p(self.hits), _p(self.cand), _p(self.slen), _p(self.flag[BUF_SDEL]))

    def sg_finish(self):
        n = self.O.orc_sg_finish(self.n_cand, _p(self.cand), _p(self.flag[BUF_SDEL]))
        self.arcs = self.cand[:n].copy()
        return n

    def export_rows(self, dst):
        rows = np.zeros((len(self.arcs), 4), "<u4")
        rows[:, 0], rows[:, 1] = self.arcs["ul"] >> 32, self.arcs["v"]
        rows[:, 2], rows[:, 3] = self.arcs["ul"] & 0xFFFFFFFF, self.arcs["oldel"]
        dst.numpy()[:rows.nbytes] = rows.view(np.uint8).reshape(-1)

    def import_rows(self, src, counts, stride):
        buf = src.numpy().view("<u4").reshape(-1, 4)
        parts = [buf[r * stride:r * stride + counts[r]] for r in range(len(counts))]
        rows = np.concatenate(parts) if parts else np.zeros((0, 4), "<u4")
        self.arcs = np.zeros(len(rows), ARC_DT)
        self.arcs["ul"] = (rows[:, 0].astype(np.uint64) << 32) | rows[:, 2].astype(np.uint64)
        self.arcs["v"], self.arcs["oldel"] = rows[:, 1], rows[:, 3]
        self._index()

    def _index(self):
        self.idx = np.zeros(2 * max(self.R, 1), "<u8")
        self.O.orc_arc_index(self.R, len(self.arcs), _p(self.arcs), _p(self.idx))

    def del_trans_range(self, opt, v0, v1):
        if not hasattr(self, "idx"):
            self._index()
        return self.O.orc_arc_del_trans_range(self.R, len(self.arcs), _p(self.arcs), _p(self.idx), _p(self.flag[BUF_SDEL]), opt.gap_fuzz, v0, v1, None)

    def flags_out(self, dst, first, count):
        dst.numpy().view("<u4")[:count] = self.arcs["oldel"][first:first + count]

    def flags_in(self, src, byte_off, first, count):
        self.arcs["oldel"][first:first + count] = src.numpy()[byte_off:byte_off + 4 * count].view("<u4")

    def cleanup(self):
        n = self.O.orc_arc_rm(len(self.arcs), _p(self.arcs), _p(self.flag[BUF_SDEL]))
        self.arcs = self.arcs[:n].copy()
        self._index()
        return n

    def symm(self):
        nm = self.O.orc_arc_del_multi(self.R, len(self.arcs), _p(self.arcs), _p(self.idx))
        if nm:
            self.cleanup()
        na = self.O.orc_arc_del_asymm(self.R, len(self.arcs), _p(self.arcs), _p(self.idx))
        if na:
            self.cleanup()
        return nm, na

    def result_arcs_squeezed(self):
        """the graph with read ids renumbered through the squeeze map (what an export applies on the device)"""
        a = self.arcs.copy()
        u, v = (a["ul"] >> 32).astype(np.int64), a["v"].astype(np.int64)
        nu = (self.map[u >> 1].astype(np.int64) << 1) | (u & 1)
        nv = (self.map[v >> 1].astype(np.int64) << 1) | (v & 1)
        a["ul"] = (nu.astype(np.uint64) << 32) | (a["ul"] & np.uint64(0xFFFFFFFF))
        a["v"] = nv.astype(np.uint32)
        return a
