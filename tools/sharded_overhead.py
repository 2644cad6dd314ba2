"""probe: cost of the Python-orchestrated sharded path on ONE rank against the C pipeline (same data, same result)"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniasm_amd as ma
from miniasm_amd.sharded import Comm, GpuBackend, run_sharded
L = ma.lib(); L.ma_set_log_path(b"/dev/null"); L.sys_init()
paf = "/tmp/so.paf"
os.system("%s -r 200000 -n 10000000 -s 1 -o %s 2>/dev/null" % (ma.PAFGEN_PATH, paf))
opt = ma.default_opt()
ing = ma.Ingest(paf, opt)
n, n_seq, max_qs = ing.n, ing.n_seq, ing.max_qs
dev = torch.from_numpy(ing.hits.view("u1").reshape(-1).copy()).cuda()
ing.free_hits()
be = GpuBackend.create(0, n_seq); ctx = be.ctx; comm = Comm()
buf, ln = C.c_void_p(0), C.c_size_t(0)
L.ma_pipeline_tail_mem.restype = C.c_int
L.ma_pipeline_tail_mem.argtypes = [C.c_void_p, C.POINTER(ma.MaOpt), C.POINTER(ma.Sdict), C.c_char_p, C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
def step_sharded():
    L.mahip_hits_adopt(ctx.h, dev.data_ptr(), n, n_seq); L.mahip_set_hints(ctx.h, max_qs)
    stats = run_sharded(be, comm, opt, n_seq)
    st = (C.c_uint32 * 4)(1, 1, stats["n_red"], 1)
    assert L.ma_pipeline_tail_mem(ctx.h, C.byref(opt), ing.d, b"ug", 100, C.byref(st), C.byref(buf), C.byref(ln)) == 0
    out = C.string_at(buf, ln.value); L.free_buf(buf); return out
def step_c():
    L.mahip_hits_adopt(ctx.h, dev.data_ptr(), n, n_seq); L.mahip_set_hints(ctx.h, max_qs)
    assert L.ma_pipeline_device_mem(ctx.h, C.byref(opt), ing.d, b"ug", 100, 0, C.byref(buf), C.byref(ln)) == 0
    out = C.string_at(buf, ln.value); L.free_buf(buf); return out
a, b = step_sharded(), step_c()
assert a == b, "outputs differ"
for name, f in (("C pipeline", step_c), ("sharded path, 1 rank", step_sharded), ("C pipeline", step_c), ("sharded path, 1 rank", step_sharded)):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print("%-22s %.3f ms/step" % (name, (time.perf_counter() - t0) * 100))
