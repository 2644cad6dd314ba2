"""CPU, world_size 2 and 3 over gloo: the product's sharded mode -- host/sharded.c (read-range shards, interval all-gathers, flag max-all-reduces, the arc
all-gather, ranged transitive reduction, del-flag all-gather, tie repair) and host/ingest_sharded.c (every rank on its own byte range of the text, name
tables merged, records routed to their owners) -- run by ranks that torch.distributed started, with gloo as the transport of every collective
(miniasm_amd/dist_transport.py -> mahip_comm_init_ext) and the CPU build of the kernels (tests/emu) underneath.  The GFA rank 0 writes must be the
reference's, byte for byte.  Until round 3 this module drove a Python restatement of the exchange sequence with an oracle-backed stand-in; now it is the C code
itself, and there is one copy of the sequence."""
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "emu")
EMU_LIB = os.path.join(EMU, "_build", "libminiasm_amd_emu.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, paf, out_path, whole):
    os.environ["MINIASM_AMD_LIB"] = EMU_LIB  # before the package is imported: the kernels' CPU build
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if whole:
        os.environ["MA_INGEST_WHOLE"] = "1"
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import miniasm_amd as ma
    from miniasm_amd.dist_transport import Transport
    L = ma.lib()
    L.ma_set_log_path(b"/dev/null")
    L.sys_init()
    ctx = ma.Ctx(0)
    tr = Transport()
    tr.attach(L, ctx.h)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    fp = libc.fopen((out_path if rank == 0 else "/dev/null").encode(), b"w")
    opt = ma.default_opt()
    L.ma_pipeline_run_rank.restype = C.c_int
    L.ma_pipeline_run_rank.argtypes = [C.c_void_p, C.POINTER(ma.MaOpt), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rc = L.ma_pipeline_run_rank(ctx.h, C.byref(opt), paf.encode(), b"ug", 100, 0, fp, 1)
    libc.fclose(fp)
    assert rc == 0
    assert tr.calls > 8, "the collectives were supposed to go through torch.distributed (%d calls)" % tr.calls
    L.mahip_comm_destroy.argtypes = [C.c_void_p]
    L.mahip_comm_destroy(ctx.h)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def emu_built(built):
    r = subprocess.run(["make", "-C", EMU, "-j8", "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    return True


@pytest.mark.parametrize("world,case,whole", [(2, "lognormal", 0), (3, "noisy", 0), (2, "ties", 0), (3, "lognormal", 1)])
def test_sharded_mode_over_gloo_matches_the_reference(world, case, whole, tmpdir_s, emu_built):
    sys.path.insert(0, HERE)
    import refapi as R
    if not R.have_ref():
        pytest.skip("oracle/_ref not built")
    extra = {"lognormal": [], "noisy": ["-L", "uniform", "-d", "0.35", "-x", "0.03"], "ties": ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"]}[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "dist_%s.paf" % case), 1201, 30000, 71, extra)  # 1201 reads: uneven shards
    out = os.path.join(tmpdir_s, "dist_%s_%d_%d.gfa" % (case, world, whole))
    mp.spawn(_worker, args=(world, _free_port(), paf, out, whole), nprocs=world, join=True)
    ref, _ = R.run_cli(R.REF_BIN, [], paf)
    got = open(out, "rb").read()
    assert got.startswith(b"S\t") and got == ref, "%d ranks over gloo: the GFA differs from the reference's" % world
