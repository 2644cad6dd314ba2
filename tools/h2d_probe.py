"""probe: host->device copy rates on the GPU box (pinned vs pageable vs staged workers)"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniasm_amd as ma
n = 640 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory(); pin.fill_(3)
pag = torch.empty(n, dtype=torch.uint8); pag.fill_(5)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best / 1e9
print("pinned  H2D %.1f GB/s" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("pageable H2D (torch) %.1f GB/s" % t(lambda: dev.copy_(pag)))
print("pinned  D2H %.1f GB/s" % t(lambda: pin.copy_(dev, non_blocking=True)))
ctx = ma.Ctx(0)
for w in (1, 2, 4, 8, 16):
    os.environ["MA_XFER_THREADS"] = str(w)
    print("staged H2D %2d workers %.1f GB/s" % (w, t(lambda: ma.lib().mahip_memcpy_h2d(ctx.h, dev.data_ptr(), pag.data_ptr(), n))))
for w in (4, 8):
    os.environ["MA_XFER_THREADS"] = str(w)
    print("staged D2H %2d workers %.1f GB/s" % (w, t(lambda: ma.lib().mahip_memcpy_d2h(ctx.h, pag.data_ptr(), dev.data_ptr(), n))))
t0 = time.perf_counter(); b = bytearray(n); t1 = time.perf_counter()
import numpy as np
a = np.frombuffer(b, dtype=np.uint8); src = pag.numpy()
t0 = time.perf_counter(); a[:] = src; t1 = time.perf_counter()
print("host memcpy 1 thread %.1f GB/s" % (n / (t1 - t0) / 1e9))
