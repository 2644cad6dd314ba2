"""miniasm_amd -- MI355X-native overlap-graph hot path of miniasm (PAF hits -> string graph -> GFA).

The product is native: hand-written HIP kernels (miniasm_amd/csrc) behind a C ABI (include/mahip.h), host C
that mirrors the reference's link interface (include/miniasm_amd.h, miniasm_amd/host) and the `miniasm`
command line (miniasm_amd/bin/miniasm).  This Python module is only the thin ctypes harness that tests and
bench.py use to drive the C ABI; it contains no compute and no fallback path: without the built library, or
without a GPU, calls fail loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "miniasm_amd")
LIB_PATH = os.environ.get("MINIASM_AMD_LIB") or os.path.join(PKG, "lib", "libminiasm_amd.so")  # the override is for kernel-variant experiments (tools/variants.sh)
CLI_PATH = os.path.join(PKG, "bin", "miniasm")
PAFGEN_PATH = os.path.join(PKG, "bin", "pafgen")

# byte layouts of the reference records (miniasm.h:29-40, asg.h:7-15), as numpy structured dtypes
HIT_DT = np.dtype([("qns", "<u8"), ("qe", "<u4"), ("tn", "<u4"), ("ts", "<u4"), ("te", "<u4"), ("mlrev", "<u4"), ("bldel", "<u4")])
SUB_DT = np.dtype([("sdel", "<u4"), ("e", "<u4")])
ARC_DT = np.dtype([("ul", "<u8"), ("v", "<u4"), ("oldel", "<u4")])
assert HIT_DT.itemsize == 32 and SUB_DT.itemsize == 8 and ARC_DT.itemsize == 16


SHARD_N_PHASES = 16
SHARD_PHASE_NAMES = ["sort", "sub#1", "x:sub0", "cut+flt+sub#2", "x:sub1", "merge+cut+contained", "x:flags", "squeeze+sg flags", "x:seq.del", "local arcs",
                     "x:arc counts", "x:arc blocks", "tie repair", "reduction (own vertices)", "x:del flags", "rank 0: cleanup+symm"]  # host/sharded.c: ma_shard_phase_name


class ShardStats(C.Structure):  # host/ma_host.h: ma_shard_stats_t
    _fields_ = [("n_rem1", C.c_uint64), ("n_rem2", C.c_uint64), ("n_hits", C.c_uint64), ("n_seq_new", C.c_uint32), ("n_arc", C.c_uint32), ("n_loc_arc", C.c_uint32),
                ("n_red", C.c_uint32), ("n_multi", C.c_uint32), ("n_asymm", C.c_uint32), ("tie_groups", C.c_uint64), ("push_conflicts", C.c_uint64), ("tie_repaired", C.c_int),
                ("n_red_local", C.c_uint32), ("reduced", C.c_int), ("have_phases", C.c_int), ("phase_ms", C.c_float * SHARD_N_PHASES), ("xchg_bytes", C.c_uint64 * SHARD_N_PHASES)]


class MaOpt(C.Structure):  # miniasm.h:12-27
    _fields_ = [("min_span", C.c_int), ("min_match", C.c_int), ("min_dp", C.c_int), ("min_iden", C.c_float),
                ("max_hang", C.c_int), ("min_ovlp", C.c_int), ("int_frac", C.c_float),
                ("gap_fuzz", C.c_int), ("n_rounds", C.c_int), ("bub_dist", C.c_int), ("max_ext", C.c_int),
                ("min_ovlp_drop_ratio", C.c_float), ("max_ovlp_drop_ratio", C.c_float), ("final_ovlp_drop_ratio", C.c_float)]


class SdSeq(C.Structure):  # sdict.h:6-9
    _fields_ = [("name", C.c_char_p), ("len", C.c_uint32), ("auxdel", C.c_uint32)]


class Sdict(C.Structure):  # sdict.h:11-15
    _fields_ = [("n_seq", C.c_uint32), ("m_seq", C.c_uint32), ("seq", C.POINTER(SdSeq)), ("h", C.c_void_p)]


class Asg(C.Structure):  # asg.h:17-23
    _fields_ = [("m_arc", C.c_uint32), ("n_arc_srt", C.c_uint32), ("arc", C.c_void_p),
                ("m_seq", C.c_uint32), ("n_seq_symm", C.c_uint32), ("seq", C.c_void_p), ("idx", C.c_void_p)]

    @property
    def n_arc(self):
        return self.n_arc_srt & 0x7FFFFFFF

    @property
    def n_seq(self):
        return self.n_seq_symm & 0x7FFFFFFF


class TieInfo(C.Structure):  # include/mahip.h: mahip_tie_info_t
    _fields_ = [("arc_tie_groups", C.c_uint64), ("arc_tie_arcs", C.c_uint64), ("push_conflicts", C.c_uint64), ("hit_ties", C.c_uint64),
                ("arc_walk", C.c_int), ("hit_walk", C.c_int), ("unrepaired", C.c_int), ("push_conflicts_seen", C.c_uint64), ("hit_walk_reads", C.c_uint64)]


class ProfRec(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double), ("alg_bytes", C.c_double)]


def build(verbose=False):
    """Compile every HIP and C source in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    cmd = ["make", "-C", ROOT, "-j8", "all"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("miniasm_amd build failed")


_lib = None


def lib():
    """The product library.  Raises if it has not been built; never substitutes anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libminiasm_amd.so is not built (run `make` or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
        L.mahip_create.restype = vp
        L.mahip_create.argtypes = [i32, vp]
        L.mahip_destroy.argtypes = [vp]
        L.mahip_strerror.restype = C.c_char_p
        L.mahip_device_count.restype = i32
        L.mahip_sync.argtypes = [vp]
        L.mahip_hits_upload.argtypes = [vp, vp, sz, u32]
        L.mahip_hits_adopt.argtypes = [vp, vp, sz, u32]
        L.mahip_set_shard.argtypes = [vp, u32, u32]
        L.mahip_set_hints.argtypes = [vp, u32]
        L.mahip_set_run_stride.argtypes = [vp, i32]
        L.mahip_set_exact_ties.argtypes = [vp, i32]
        L.mahip_tie_stats.argtypes = [vp, C.POINTER(TieInfo)]
        L.mahip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
        L.mahip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
        L.mahip_paf_release.argtypes = [vp]
        L.mahip_hits_sorted_runs.restype = C.c_uint64
        L.mahip_hits_sorted_runs.argtypes = [vp]
        L.mahip_first_launch.argtypes = [vp]
        L.mahip_paf_load_fd.argtypes = [vp, i32, sz]
        L.mahip_paf_load_mem.argtypes = [vp, vp, sz]
        L.mahip_hits_raw_download.argtypes = [vp, vp]
        L.mahip_hits_sort.argtypes = [vp]
        L.mahip_hits_index.argtypes = [vp]
        L.mahip_hits_sub.argtypes = [vp, i32, C.c_float, i32, i32, C.POINTER(sz)]
        L.mahip_hits_cut.argtypes = [vp, i32, i32, C.POINTER(sz)]
        L.mahip_hits_flt.argtypes = [vp, i32, i32, i32, C.POINTER(sz), C.POINTER(C.c_float)]
        L.mahip_sub_merge.argtypes = [vp]
        L.mahip_hits_contained.argtypes = [vp, C.POINTER(MaOpt), vp, C.POINTER(u32), C.POINTER(sz)]
        L.mahip_sub_upload.argtypes = [vp, i32, vp, sz]
        L.mahip_sub_download.argtypes = [vp, i32, vp, i32]
        L.mahip_seqdel_download.argtypes = [vp, vp]
        L.mahip_map_download.argtypes = [vp, vp]
        L.mahip_hits_live.restype = sz
        L.mahip_hits_live.argtypes = [vp]
        L.mahip_hits_download.argtypes = [vp, vp, C.POINTER(sz)]
        L.mahip_sg_gen.argtypes = [vp, C.POINTER(MaOpt), i32, vp, vp, C.POINTER(u32)]
        L.mahip_asg_upload.argtypes = [vp, C.POINTER(Asg)]
        L.mahip_asg_del_trans.argtypes = [vp, i32, C.POINTER(u32)]
        L.mahip_asg_symm.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
        L.mahip_asg_del_short.argtypes = [vp, C.c_float, C.POINTER(u32)]
        L.mahip_asg_n_arc.restype = u32
        L.mahip_asg_n_arc.argtypes = [vp]
        L.mahip_asg_download.argtypes = [vp, C.POINTER(Asg)]
        L.mahip_prof_enable.argtypes = [vp, i32]
        L.mahip_prof_reset.argtypes = [vp]
        L.mahip_prof_get.argtypes = [vp, C.POINTER(ProfRec), i32]
        L.mahip_mem_bytes.restype = sz
        L.mahip_mem_bytes.argtypes = [vp]
        L.ma_opt_init.argtypes = [C.POINTER(MaOpt)]
        L.sd_init.restype = C.POINTER(Sdict)
        L.sd_destroy.argtypes = [C.POINTER(Sdict)]
        L.ma_hit_ingest.restype = vp
        L.ma_hit_ingest.argtypes = [C.c_char_p, i32, i32, C.POINTER(Sdict), C.POINTER(sz), i32, vp]
        L.ma_pipeline_device_mem.restype = i32
        L.ma_pipeline_device_mem.argtypes = [vp, C.POINTER(MaOpt), C.POINTER(Sdict), C.c_char_p, i32, i32, C.POINTER(vp), C.POINTER(sz)]
        L.ma_set_log_path.argtypes = [C.c_char_p]
        L.sys_init.argtypes = []
        L.free_buf = C.CDLL(None).free
        L.free_buf.argtypes = [vp]
        _lib = L
    return _lib


def default_opt():
    o = MaOpt()
    lib().ma_opt_init(C.byref(o))
    return o


class GpuError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise GpuError("%s failed: %s" % (what, lib().mahip_strerror().decode()))


class Ctx:
    """One GPU context (include/mahip.h).  Construction fails loudly when no GPU is usable."""

    def __init__(self, device=0, stream=None):
        L = lib()
        self.h = L.mahip_create(device, stream)
        if not self.h:
            raise GpuError(L.mahip_strerror().decode())

    def close(self):
        if self.h:
            lib().mahip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exact_ties(self, mode):
        lib().mahip_set_exact_ties(self.h, mode)

    def tie_stats(self):
        t = TieInfo()
        lib().mahip_tie_stats(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in TieInfo._fields_}

    # ---- hits
    def hits_upload(self, hits, n_seq):
        hits = np.ascontiguousarray(hits, dtype=HIT_DT)
        self._keep = hits
        _chk(lib().mahip_hits_upload(self.h, hits.ctypes.data, len(hits), n_seq), "hits_upload")
        _chk(lib().mahip_sync(self.h), "sync")

    def set_run_stride(self, stride):
        """hint for the sort (include/mahip.h): 2 = records and mirrors side by side, 1 = no mirrors, 0 = unknown; describes one upload"""
        _chk(lib().mahip_set_run_stride(self.h, int(stride)), "set_run_stride")

    def hits_adopt(self, dptr, n, n_seq):
        _chk(lib().mahip_hits_adopt(self.h, dptr, n, n_seq), "hits_adopt")

    def sort(self):
        _chk(lib().mahip_hits_sort(self.h), "hits_sort")

    def sorted_runs(self):
        """elements of the last sort if it sorted RUNS of records (hits.hip), else 0: lets a test say which path it took"""
        return int(lib().mahip_hits_sorted_runs(self.h))

    def index(self):
        _chk(lib().mahip_hits_index(self.h), "hits_index")

    def sub(self, min_dp, min_iden, end_clip, slot=0):
        n = C.c_size_t(0)
        _chk(lib().mahip_hits_sub(self.h, min_dp, min_iden, end_clip, slot, C.byref(n)), "hits_sub")
        return n.value

    def cut(self, slot, min_span):
        n = C.c_size_t(0)
        _chk(lib().mahip_hits_cut(self.h, slot, min_span, C.byref(n)), "hits_cut")
        return n.value

    def flt(self, slot, max_hang, min_ovlp):
        n = C.c_size_t(0)
        cov = C.c_float(0)
        _chk(lib().mahip_hits_flt(self.h, slot, max_hang, min_ovlp, C.byref(n), C.byref(cov)), "hits_flt")
        return n.value, cov.value

    def sub_merge(self):
        _chk(lib().mahip_sub_merge(self.h), "sub_merge")

    def contained(self, opt, seq_del=None):
        n = C.c_size_t(0)
        r = C.c_uint32(0)
        p = seq_del.ctypes.data if seq_del is not None else None
        _chk(lib().mahip_hits_contained(self.h, C.byref(opt), p, C.byref(r), C.byref(n)), "hits_contained")
        return r.value, n.value

    def sub_upload(self, slot, sub):
        sub = np.ascontiguousarray(sub, dtype=SUB_DT)
        _chk(lib().mahip_sub_upload(self.h, slot, sub.ctypes.data, len(sub)), "sub_upload")

    def sub_download(self, slot, n, squeezed=False):
        out = np.zeros(n, dtype=SUB_DT)
        _chk(lib().mahip_sub_download(self.h, slot, out.ctypes.data, 1 if squeezed else 0), "sub_download")
        return out

    def seqdel_download(self, n):
        out = np.zeros(n, dtype=np.uint8)
        _chk(lib().mahip_seqdel_download(self.h, out.ctypes.data), "seqdel_download")
        return out

    def map_download(self, n):
        out = np.zeros(n, dtype=np.int32)
        _chk(lib().mahip_map_download(self.h, out.ctypes.data), "map_download")
        return out

    def hits_download(self):
        n = lib().mahip_hits_live(self.h)
        out = np.zeros(max(n, 1), dtype=HIT_DT)
        m = C.c_size_t(0)
        _chk(lib().mahip_hits_download(self.h, out.ctypes.data, C.byref(m)), "hits_download")
        return out[:m.value]

    # ---- graph
    def sg_gen(self, opt, use_sub=True, seq_len=None, seq_del=None):
        n = C.c_uint32(0)
        pl = seq_len.ctypes.data if seq_len is not None else None
        pd = seq_del.ctypes.data if seq_del is not None else None
        _chk(lib().mahip_sg_gen(self.h, C.byref(opt), 1 if use_sub else 0, pl, pd, C.byref(n)), "sg_gen")
        return n.value

    def del_trans(self, fuzz):
        n = C.c_uint32(0)
        _chk(lib().mahip_asg_del_trans(self.h, fuzz, C.byref(n)), "asg_del_trans")
        return n.value

    def symm(self):
        a, b = C.c_uint32(0), C.c_uint32(0)
        _chk(lib().mahip_asg_symm(self.h, C.byref(a), C.byref(b)), "asg_symm")
        return a.value, b.value

    def del_short(self, ratio):
        n = C.c_uint32(0)
        _chk(lib().mahip_asg_del_short(self.h, ratio, C.byref(n)), "asg_del_short")
        return n.value

    def asg_download(self):
        """-> (arcs[ARC_DT], seq[u4], idx[u8]) copies; frees the C arrays."""
        g = Asg()
        _chk(lib().mahip_asg_download(self.h, C.byref(g)), "asg_download")
        na, ns = g.n_arc, g.n_seq
        arcs = np.frombuffer(C.string_at(g.arc, na * 16), dtype=ARC_DT).copy() if na else np.zeros(0, ARC_DT)
        seq = np.frombuffer(C.string_at(g.seq, ns * 4), dtype="<u4").copy() if ns else np.zeros(0, "<u4")
        idx = np.frombuffer(C.string_at(g.idx, ns * 16), dtype="<u8").copy() if ns else np.zeros(0, "<u8")
        for p in (g.arc, g.seq, g.idx):
            lib().free_buf(p)
        return arcs, seq, idx

    # ---- instrumentation
    def prof_enable(self, on=True):
        lib().mahip_prof_enable(self.h, 1 if on else 0)

    def prof_reset(self):
        lib().mahip_prof_reset(self.h)

    def prof_get(self):
        buf = (ProfRec * 64)()
        n = lib().mahip_prof_get(self.h, buf, 64)
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, alg_bytes=buf[i].alg_bytes) for i in range(min(n, 64))]

    def mem_bytes(self):
        return lib().mahip_mem_bytes(self.h)


class Ingest:
    """Host ingest of a PAF file (reference hit.c:70-101, everything before the sort): hits + dictionary."""

    def __init__(self, fn, opt=None, bi_dir=True):
        L = lib()
        opt = opt or default_opt()
        self.d = L.sd_init()
        n = C.c_size_t(0)
        p = L.ma_hit_ingest(fn.encode(), opt.min_span, opt.min_match, self.d, C.byref(n), 1 if bi_dir else 0, None)
        self.n = n.value
        L.ma_ingest_max_qs.restype = C.c_uint32
        self.max_qs = L.ma_ingest_max_qs()  # exact bound of the query starts: hint for the device sort
        self._p = p  # malloc'ed by the library; self.hits is a zero-copy view of it (no second copy of multi-GB arrays)
        if self.n:
            raw = (C.c_uint8 * (self.n * 32)).from_address(p)
            self.hits = np.frombuffer(raw, dtype=HIT_DT)
        else:
            self.hits = np.zeros(0, HIT_DT)
        self.n_seq = self.d.contents.n_seq

    def names(self):
        d = self.d.contents
        return [d.seq[i].name.decode() for i in range(d.n_seq)]

    def lens(self):
        d = self.d.contents
        return np.array([d.seq[i].len for i in range(d.n_seq)], dtype=np.uint32)

    def free_hits(self):
        """release the host copy of the hit records (the dictionary stays)"""
        self.hits = np.zeros(0, HIT_DT)
        if self._p:
            lib().free_buf(self._p)
            self._p = None

    def close(self):
        self.free_hits()
        if self.d:
            lib().sd_destroy(self.d)
            self.d = None


class GpuIngest:
    """Device-side ingest (csrc/paf.hip through host/ingest_gpu.c): the records stay in `ctx`; the dictionary comes back.
    Same attributes as Ingest; `hits` downloads the unsorted records (tests)."""

    def __init__(self, ctx, fn, opt=None, bi_dir=True):
        L = lib()
        opt = opt or default_opt()
        self.ctx = ctx
        self.d = L.sd_init()
        n = C.c_size_t(0)
        L.ma_hit_ingest_gpu.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(Sdict), C.POINTER(C.c_size_t), C.c_int]
        rc = L.ma_hit_ingest_gpu(ctx.h, fn.encode(), opt.min_span, opt.min_match, self.d, C.byref(n), 1 if bi_dir else 0)
        if rc != 0:
            raise OSError("cannot open %s" % fn)
        self.n = n.value
        self.n_seq = self.d.contents.n_seq

    @property
    def hits(self):
        out = np.zeros(self.n, dtype=HIT_DT)
        L = lib()
        L.mahip_hits_raw_download.argtypes = [C.c_void_p, C.c_void_p]
        _chk(L.mahip_hits_raw_download(self.ctx.h, out.ctypes.data), "hits_raw_download")
        return out

    names = Ingest.names
    lens = Ingest.lens

    def close(self):
        if self.d:
            lib().sd_destroy(self.d)
            self.d = None


def run_resident(ctx, opt, ingest, outfmt="ug", stage=100, flags=0):
    """Everything after ingest with the hits resident in HBM (pipeline.c:ma_pipeline_device); returns the output text."""
    L = lib()
    buf = C.c_void_p(0)
    ln = C.c_size_t(0)
    rc = L.ma_pipeline_device_mem(ctx.h, C.byref(opt), ingest.d, outfmt.encode(), stage, flags, C.byref(buf), C.byref(ln))
    if rc != 0:
        raise GpuError("pipeline failed: %s" % L.mahip_strerror().decode())
    out = C.string_at(buf, ln.value)
    L.free_buf(buf)
    return out


def run_resident_handoff(ctx, ctx2, opt, ingest, outfmt="ug", stage=100, flags=0):
    """The same job split over two contexts of one device (include/mahip.h: mahip_tail_handoff): hit passes, graph and reduction on ctx,
    cleaners + unitigs + downloads on ctx2 -- what a caller with a stream of inputs overlaps with the next input's hit passes."""
    L = lib()
    vp = C.c_void_p
    L.ma_pipeline_head.restype = C.c_int
    L.ma_pipeline_head.argtypes = [vp, C.POINTER(MaOpt), C.POINTER(Sdict), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32 * 4)]
    L.ma_pipeline_tail_mem.restype = C.c_int
    L.ma_pipeline_tail_mem.argtypes = [vp, C.POINTER(MaOpt), C.POINTER(Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mahip_tail_handoff.argtypes = [vp, vp]
    st = (C.c_uint32 * 4)(0, 0, 0, 0)
    if L.ma_pipeline_head(ctx.h, C.byref(opt), ingest.d, outfmt.encode(), stage, flags, C.byref(st)) != 0:
        raise GpuError("pipeline head failed: %s" % L.mahip_strerror().decode())
    _chk(L.mahip_tail_handoff(ctx.h, ctx2.h), "tail_handoff")
    buf, ln = vp(0), C.c_size_t(0)
    if L.ma_pipeline_tail_mem(ctx2.h, C.byref(opt), ingest.d, outfmt.encode(), stage, C.byref(st), C.byref(buf), C.byref(ln)) != 0:
        raise GpuError("pipeline tail failed: %s" % L.mahip_strerror().decode())
    out = C.string_at(buf, ln.value)
    L.free_buf(buf)
    return out
