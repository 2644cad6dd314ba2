"""TEST INFRASTRUCTURE: the handful of torch calls bench.py makes (device buffers, synchronize), backed by numpy, so that bench.py's own control
flow -- Workload, Runner, the worker thread, --tail-ctx, the JSON line -- can be executed on the CPU build of the kernels (tests/emu), where
device pointers are host pointers.  Only tests/test_emu_suite.py puts this directory on PYTHONPATH."""
import numpy as np

uint8 = np.uint8
float64 = np.float64
__version__ = "0+emu"


class _Tensor:
    def __init__(self, a):
        self.a = a

    def data_ptr(self):
        return self.a.ctypes.data

    def __getitem__(self, i):
        return self.a[i]


def empty(n, dtype=None, device=None):
    return _Tensor(np.zeros(int(n), dtype=dtype or np.uint8))


def tensor(x, dtype=None):
    return _Tensor(np.array(x, dtype=dtype))


class cuda:
    is_available = staticmethod(lambda: True)
    set_device = staticmethod(lambda d: None)
    synchronize = staticmethod(lambda: None)
