"""The measurement hooks of csrc/diag.hip (tools/pmc_calibrate.py): every access pattern runs, reports the byte count it has by construction and a time."""
import ctypes as C

import pytest

import miniasm_amd as ma

pytestmark = pytest.mark.gpu


def test_every_diag_pattern_runs_and_reports_its_bytes():
    L = ma.lib()
    L.mahip_diag_name.restype = C.c_char_p
    L.mahip_diag_run.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ctx = ma.Ctx(0)
    n = L.mahip_diag_patterns()
    assert n >= 15
    size = 2 << 20
    for p in range(n):
        name = L.mahip_diag_name(p).decode()
        ms, mv = C.c_double(-1), C.c_double(-1)
        assert L.mahip_diag_run(ctx.h, p, size, 1, C.byref(ms), C.byref(mv)) == 0, name
        assert ms.value >= 0 and mv.value >= size / 2, (name, ms.value, mv.value)
        if name.startswith("copy") or name.startswith("scatter"):
            assert mv.value == 2 * size, name
    assert L.mahip_diag_name(n) is None
    assert L.mahip_diag_run(ctx.h, n, size, 1, None, None) != 0  # no such pattern
    ctx.close()
