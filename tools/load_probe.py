"""probe: file -> HBM loader rate (mahip_paf_load_fd), cold and warm, and the parse time split by kernel"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniasm_amd as ma
paf = sys.argv[1]
L = ma.lib()
L.mahip_paf_load_fd.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
class Info(C.Structure):
    _fields_ = [("n_lines", C.c_uint64), ("n_records", C.c_uint64), ("n_stored_lines", C.c_uint64), ("n_hits", C.c_uint64), ("name_bytes", C.c_uint64), ("n_seq", C.c_uint32), ("max_qs", C.c_uint32)]
L.mahip_paf_parse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Info)]
t0 = time.perf_counter(); ctx = ma.Ctx(0); print("ctx %.3f s" % (time.perf_counter() - t0))
size = os.path.getsize(paf)
fd = os.open(paf, os.O_RDONLY)
for w in (None, "4", "8", "16", "8"):
    if w: os.environ["MA_XFER_THREADS"] = w
    t0 = time.perf_counter(); rc = L.mahip_paf_load_fd(ctx.h, fd, size); dt = time.perf_counter() - t0
    print("load_fd workers=%s rc=%d %.3f s  %.1f GB/s" % (w, rc, dt, size / dt / 1e9))
opt = ma.default_opt()
info = Info()
L.mahip_prof_enable(ctx.h, 1)
for r in range(3):
    L.mahip_prof_reset(ctx.h)
    t0 = time.perf_counter(); rc = L.mahip_paf_parse(ctx.h, opt.min_span, opt.min_match, 1, C.byref(info)); dt = time.perf_counter() - t0
    print("parse rc=%d %.2f ms  lines %d records %d hits %d reads %d" % (rc, dt * 1e3, info.n_lines, info.n_records, info.n_hits, info.n_seq))
recs = (ma.ProfRec * 64)()
n = L.mahip_prof_get(ctx.h, recs, 64)
for i in range(n):
    r = recs[i]
    print("  %-22s x%d  %.3f ms  alg %.0f MB -> %.0f GB/s" % (r.name.decode(), r.launches, r.total_ms / max(r.launches, 1), r.alg_bytes / max(r.launches, 1) / 1e6, r.alg_bytes / max(r.total_ms, 1e-9) / 1e6))
