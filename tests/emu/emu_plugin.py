"""pytest plugin (TEST INFRASTRUCTURE): points the ctypes harness at tests/emu/_build/libminiasm_amd_emu.so -- the product's
kernel sources compiled for the CPU against the fiber-based HIP stand-in -- so that `-m gpu` test cases can run where there
is no GPU.  Loaded only by tests/test_emu_suite.py (`-p emu_plugin`); the product never sees it."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.environ.get("MA_EMU_BUILD", "_build")  # another build directory of tests/emu/Makefile (B=..., EXTRA=-DEXP_...): kernel variants
EMU_LIB = os.path.join(HERE, BUILD, "libminiasm_amd_emu.so")
EMU_CLI = os.path.join(HERE, BUILD, "miniasm")

os.environ["MINIASM_AMD_LIB"] = EMU_LIB
os.environ.setdefault("MA_COMM", "shm")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import miniasm_amd as ma  # noqa: E402

assert ma.LIB_PATH == EMU_LIB
ma.CLI_PATH = EMU_CLI
ma.IS_EMU = True

sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
import refapi as R  # noqa: E402

R.DROPIN_BIN = os.path.join(HERE, BUILD, "miniasm_dropin")
