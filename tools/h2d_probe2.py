import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniasm_amd as ma
n = 640 << 20
torch.cuda.init(); torch.cuda.synchronize()
def now(): return time.perf_counter()
t0 = now(); dev = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); print("alloc dev %.3f" % (now() - t0))
t0 = now(); dev.zero_(); torch.cuda.synchronize(); print("zero dev (first touch) %.3f" % (now() - t0))
t0 = now(); dev.zero_(); torch.cuda.synchronize(); print("zero dev again %.3f" % (now() - t0))
src = np.empty(n, dtype=np.uint8); t0 = now(); src[:] = 7; print("host first touch %.3f" % (now() - t0))
t0 = now(); ctx = ma.Ctx(0); print("ctx %.3f" % (now() - t0))
for r in range(3):
    t0 = now(); ma.lib().mahip_memcpy_h2d(ctx.h, dev.data_ptr(), src.ctypes.data, n); print("staged rep %d %.3f s" % (r, now() - t0))
src2 = np.empty(n, dtype=np.uint8); src2[:] = 9
t0 = now(); ma.lib().mahip_memcpy_h2d(ctx.h, dev.data_ptr(), src2.ctypes.data, n); print("staged new src %.3f s" % (now() - t0))
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
t0 = now(); ma.lib().mahip_memcpy_h2d(ctx.h, dev2.data_ptr(), src2.ctypes.data, n); print("staged new dst (untouched) %.3f s" % (now() - t0))
t0 = now(); ma.lib().mahip_memcpy_h2d(ctx.h, dev2.data_ptr(), src2.ctypes.data, n); print("staged same dst again %.3f s" % (now() - t0))
