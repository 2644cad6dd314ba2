"""GPU: the sharded multi-GPU mode on real hardware.  One box has one GPU, and RCCL refuses two ranks on one device, so N ranks run as N processes
that share the GPU and exchange through the shared-memory double of the collectives (MA_COMM=shm).  Everything else is the production path: host/sharded.c,
host/ingest_sharded.c, mahip_set_shard, the split contained / sg passes, arc row import/export, ranged transitive reduction.  (Until round 3 this module
also ran N virtual ranks as threads through a Python copy of the exchange sequence; that copy is gone -- tests/test_dist_gloo.py runs the C code over gloo.)"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import miniasm_amd as ma
import refapi as R
import stages as ST

pytestmark = pytest.mark.gpu


# ---- the product path: orchestration in C (host/sharded.c), collectives from C (csrc/comm.hip) ----
@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["lognormal", "noisy", "fixed"])
def test_cli_on_n_ranks_matches_one_gpu_and_reference(world, case, tmpdir_s):
    """MA_GPUS=N miniasm: one process per rank, read-range shards, the C orchestration end to end.  This box has one GPU, where RCCL
    refuses several ranks, so the collectives run through the host-staged shared-memory double (MA_COMM=shm); everything else --
    shard bookkeeping, exchange points, buffer layout, rank-0 tail -- is the production code."""
    import subprocess
    extra = {"lognormal": [], "noisy": ["-L", "uniform", "-d", "0.35", "-x", "0.03"], "fixed": ["-L", "fixed"]}[case]
    paf = R.pafgen(os.path.join(tmpdir_s, "shc_%s.paf" % case), 3001, 80000, 81, extra)
    one, _ = R.run_cli(ma.CLI_PATH, [], paf)
    env = dict(os.environ, MA_GPUS=str(world), MA_COMM="shm")
    r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == one, "N-rank GFA differs from the single-GPU GFA"
    ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
    if R.arc_tie_groups(ref_sg) == 0:
        ref, _ = R.run_cli(R.REF_BIN, [], paf)
        assert r.stdout == ref
    r = subprocess.run([ma.CLI_PATH, "-p", "sg", paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout == R.run_cli(ma.CLI_PATH, ["-p", "sg"], paf)[0]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("world", [2, 3])
def test_cli_on_n_ranks_reproduces_the_reference_tie_order(world, tmpdir_s):
    """tie-rich input on shards: the census runs on the merged graph, the ranks restore the reference's hit order for their own pushed arcs
    (every rank holds the whole input) and rebuild the reference's arc order from the global push sequence -- GFA and string-graph dump
    byte-identical to the reference's, as on one GPU"""
    import subprocess
    paf = R.pafgen(os.path.join(tmpdir_s, "shc_ties.paf"), 3000, 80000, 5, ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"])
    ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
    assert R.arc_tie_groups(ref_sg) >= 5
    env = dict(os.environ, MA_GPUS=str(world), MA_COMM="shm")
    env.pop("MA_EXACT_TIES", None)
    for args in ([], ["-p", "sg", "-S6"], ["-p", "sg"]):
        ref, _ = R.run_cli(R.REF_BIN, args, paf)
        r = subprocess.run([ma.CLI_PATH] + args + [paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert b"left in the stable order" not in r.stderr
        assert r.stdout == ref, "%d ranks, %s: bytes differ from the reference" % (world, " ".join(args))


@pytest.mark.parametrize("tail_ctx", [0, 1])
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_that_hold_only_their_own_records(world, tail_ctx, tmpdir_s, monkeypatch):
    """the shape of `bench.py --gpus N` without torch: every rank is a process that holds ONLY the records of its read range, three steps over
    the shared-memory double, rank 0 finishes each batch on its own context or (tail_ctx) on a second one (mahip_tail_handoff); every step's GFA
    equals the single-context run and the reference's"""
    import subprocess
    import sys
    paf = R.pafgen(os.path.join(tmpdir_s, "own_%d.paf" % world), 2500, 70000, 55, ["-L", "uniform", "-d", "0.3"])
    out = os.path.join(tmpdir_s, "own_%d_%d.out" % (world, tail_ctx))
    # tail_ctx runs also use read ranges balanced by hit count (unequal ranges: the all-gathers' slots are as long as the longest one)
    env = dict(os.environ, MA_WORKER_EMU="1" if getattr(ma, "IS_EMU", False) else "0", MA_WORKER_BALANCE="1" if tail_ctx else "0")
    env.pop("MA_GPUS", None)
    name = "ma_own_%d_%d_%d" % (os.getpid(), world, tail_ctx)
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_step_worker.py"), paf, str(r), str(world), name, str(tail_ctx), out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    for p in procs:
        try:
            _, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err.decode()[-2000:]
    data = open(out, "rb").read()
    assert data.startswith(b"OK\n"), "a step's GFA differs from the single-context run"
    if R.have_ref():
        ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
        if R.arc_tie_groups(ref_sg) == 0:
            assert data[3:] == R.run_cli(R.REF_BIN, [], paf)[0]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_with_their_own_records_restore_the_reference_tie_order(world, tmpdir_s):
    """own-records shards on a tie-rich input: each rank knows where its records stood in the input (mahip_hits_set_positions), the ranks put the hit keys
    of the whole input together, every rank runs the walk (hit.c:19-22 is a function of ALL records) and orders its own pushed arcs by it; the GFA of every
    step is the reference's, byte for byte -- without positions the same run reports the groups as unrepaired"""
    import subprocess
    import sys
    paf = R.pafgen(os.path.join(tmpdir_s, "own_ties_%d.paf" % world), 3000, 80000, 5, ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"])
    ref_sg, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
    assert R.arc_tie_groups(ref_sg) >= 5
    out = os.path.join(tmpdir_s, "own_ties_%d.out" % world)
    env = dict(os.environ, MA_WORKER_EMU="1" if getattr(ma, "IS_EMU", False) else "0", MA_WORKER_BALANCE="1" if world == 3 else "0", MA_WORKER_POS="1")
    env.pop("MA_GPUS", None)
    name = "ma_ownt_%d_%d" % (os.getpid(), world)
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_step_worker.py"), paf, str(r), str(world), name, "0", out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    for p in procs:
        try:
            _, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err.decode()[-2000:]
    data = open(out, "rb").read()
    assert data.startswith(b"OK\n"), "a step's GFA differs from the single-context run"
    assert data[3:] == R.run_cli(R.REF_BIN, [], paf)[0]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("world", [2, 3, 5])
def test_every_rank_ingests_its_own_byte_range(world, tmpdir_s):
    """round-3 review, What's missing #2 (ingest routing option B): `MA_GPUS=N miniasm plain.paf` no longer parses the whole text on every rank.
    Rank g loads the bytes [g S/N, (g+1) S/N) cut at line starts, the ranks merge the name tables of their ranges into the reference's first-appearance ids
    and send every record to the owner of its query read (host/ingest_sharded.c, csrc/paf.hip, csrc/hits.hip: mahip_hits_route).  Checked here: the byte ranges
    tile the file, the output is the reference's byte for byte -- also where the reference's sequential reader carries state across a range border (a name first
    seen as a TARGET in an earlier range, 10-column lines that inherit `bl` from a line of an earlier range, CR LF, an unterminated last line, tied sort keys) --
    and MA_INGEST_WHOLE=1 (every rank parses everything: round 3) and gzip input (falls back to it) give the same bytes."""
    import gzip
    import random
    import re
    import subprocess
    base = R.pafgen(os.path.join(tmpdir_s, "ing_%d.paf" % world), 2500, 60000, 91 + world, ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"])
    rnd = random.Random(world)
    lines = open(base, "rb").read().split(b"\n")[:-1]
    out = []
    for k, ln in enumerate(lines):  # damage that crosses range borders: runs of 10-column lines (stale bl), CR LF, short lines
        f = ln.split(b"\t")
        if rnd.random() < 0.25:
            ln = b"\t".join(f[:10])
        elif rnd.random() < 0.02:
            ln = b"\t".join(f[:rnd.randint(1, 9)])
        if rnd.random() < 0.05:
            ln += b"\r"
        out.append(ln)
    paf = os.path.join(tmpdir_s, "ing_%d_damaged.paf" % world)
    with open(paf, "wb") as fo:
        fo.write(b"\n".join(out))  # no newline behind the last line
    size = os.path.getsize(paf)
    ref, _ = R.run_cli(R.REF_BIN, [], paf)
    env = dict(os.environ, MA_GPUS=str(world), MA_COMM="shm", MA_PIPE_TIMING="1")
    r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == ref, "N-rank GFA (own byte ranges) differs from the reference's"
    spans = sorted((int(a), int(b)) for a, b in re.findall(r"rank \d+ of \d+: bytes \[(\d+), (\d+)\) of %d" % size, r.stderr.decode()))
    assert len(spans) == world and spans[0][0] == 0 and spans[-1][1] == size and all(spans[k][1] == spans[k + 1][0] for k in range(world - 1)), spans
    assert max(b - a for a, b in spans) < size / world * 1.5 + 4096, "a rank loaded far more than its share: %r" % spans
    for sg_args in (["-p", "sg"], ["-p", "sg", "-S6"]):
        r2 = subprocess.run([ma.CLI_PATH] + sg_args + [paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
        assert r2.returncode == 0 and r2.stdout == R.run_cli(R.REF_BIN, sg_args, paf)[0], sg_args
    if world == 2:  # the shared-memory double moves a rank's pieces a slot's worth at a time (BASELINE configs[3] on two ranks: 3.2 GB from each through 1 GB slots): here with 512 KB slots
        small = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, MA_SHM_SLOT_LOG2="19"), timeout=900)
        assert small.returncode == 0 and small.stdout == ref, small.stderr.decode()[-500:]
    whole = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, MA_INGEST_WHOLE="1"), timeout=900)
    assert whole.returncode == 0 and whole.stdout == ref and b"bytes [" not in whole.stderr
    with open(paf, "rb") as fi, gzip.open(paf + ".gz", "wb") as fo:
        fo.write(fi.read())
    gz = subprocess.run([ma.CLI_PATH, paf + ".gz"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert gz.returncode == 0 and gz.stdout == ref


@pytest.mark.parametrize("who", ["one_gpu", "child", "parent"])
def test_cli_on_n_ranks_never_leaves_a_rank_waiting(who, tmpdir_s):
    """no rank is ever left waiting: a request the sharded head does not serve (hit dumps, early -S stages, -1 / -2) is decided before the ranks
    exist and runs on one GPU with the same output; when one rank dies in the middle, the others are not left in a collective (children die
    with the parent, the parent stops when a child ends abnormally)"""
    import subprocess
    import time
    if who != "one_gpu" and not getattr(ma, "IS_EMU", False) and os.environ.get("MA_TEST_KILL") != "1":
        pytest.skip("kills ranks in the middle of a run: host-side process handling, exercised on the CPU build; MA_TEST_KILL=1 runs it on the GPU")
    paf = R.pafgen(os.path.join(tmpdir_s, "shc_fail.paf"), 800, 12000, 7, [])
    env = dict(os.environ, MA_GPUS="3", MA_COMM="shm")
    if who == "one_gpu":
        for args in (["-p", "paf"], ["-p", "sg", "-S3"], ["-1"], ["-2", "-p", "sg"], ["-p", "bed"]):
            r = subprocess.run([ma.CLI_PATH] + args + [paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
            assert r.returncode == 0 and b"runs on one GPU" in r.stderr
            assert r.stdout == R.run_cli(ma.CLI_PATH, args, paf)[0]
    else:
        env["MA_TEST_FAIL_RANK"] = "1" if who == "child" else "0"
        t0 = time.time()
        r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
        assert r.returncode != 0 and time.time() - t0 < 60
    time.sleep(0.5)
    left = subprocess.run(["pgrep", "-f", paf], stdout=subprocess.PIPE).stdout.split()
    assert not left, "ranks still alive: %r" % left


@pytest.mark.skipif(ma.lib().mahip_device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_cli_on_two_gpus_over_rccl(tmpdir_s):
    import subprocess
    paf = R.pafgen(os.path.join(tmpdir_s, "shc_rccl.paf"), 6000, 200000, 83, ["-L", "uniform", "-d", "0.3", "-x", "0.03"])
    one, _ = R.run_cli(ma.CLI_PATH, [], paf)
    env = dict(os.environ, MA_GPUS="2")
    env.pop("MA_COMM", None)
    r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == one


@pytest.mark.skipif(getattr(ma, "IS_EMU", False), reason="RCCL needs a GPU")
def test_cli_on_one_rank_really_calls_rccl(tmpdir_s):
    """MA_GPUS=1 MA_RCCL_ONE_RANK=1: the sharded runner (host/sharded.c) on a ONE-rank RCCL communicator with the one-rank short cuts off -- every
    ncclAllGather / ncclAllReduce of the step is issued on the context's stream (symbol binding, datatypes, in-place reduction, ordering against the
    kernels around them), which is as much of the RCCL path as a box with one GPU can run.  Output = the plain single-GPU run's, tie-rich input included."""
    import subprocess
    for k, extra in enumerate((["-L", "uniform", "-d", "0.3", "-x", "0.03"], ["-q", "16", "-L", "uniform", "-d", "0.3", "-x", "0.03"])):
        paf = R.pafgen(os.path.join(tmpdir_s, "shc_rccl1_%d.paf" % k), 5000, 150000, 84 + k, extra)
        one, _ = R.run_cli(ma.CLI_PATH, [], paf)
        env = dict(os.environ, MA_GPUS="1", MA_RCCL_ONE_RANK="1")
        env.pop("MA_COMM", None)
        r = subprocess.run([ma.CLI_PATH, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout == one


def test_rccl_loads_and_initialises_a_one_rank_communicator():
    """librccl is opened at run time (dlopen); a one-rank communicator comes up on this GPU and the collectives degenerate to copies"""
    L = ma.lib()
    L.mahip_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.mahip_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mahip_comm_destroy.argtypes = [C.c_void_p]
    L.mahip_comm_world.argtypes = [C.c_void_p]
    ctx = ma.Ctx(0)
    idb = C.create_string_buffer(128)
    ma._chk(L.mahip_comm_unique_id(idb), "comm_unique_id")
    ma._chk(L.mahip_comm_init(ctx.h, idb.raw, 0, 1), "comm_init")
    assert L.mahip_comm_world(ctx.h) == 1
    a = torch.arange(1000, dtype=torch.uint8, device="cuda")
    b = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    ma._chk(L.mahip_comm_all_gather(ctx.h, a.data_ptr(), b.data_ptr(), 1000), "all_gather")
    ma._chk(L.mahip_sync(ctx.h), "sync")
    assert torch.equal(a, b)
    L.mahip_comm_destroy(ctx.h)
    ctx.close()


def test_a_table_of_read_ranges_describes_one_upload():
    """round-4 review, What's weak #9: a table of read ranges (mahip_set_shard_bounds / mahip_hits_balance) used to survive a new input with the same number
    of reads, so a second input of equal read count silently got the stale ranges.  Like the hints and the positions it now describes ONE upload: every
    upload / adopt forgets it, the caller installs it again behind the records (bench.py and tests/shard_step_worker.py do)."""
    L = ma.lib()
    L.mahip_set_shard_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    L.mahip_shard_bounds.restype = C.POINTER(C.c_uint32)
    L.mahip_shard_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    ctx = ma.Ctx(0)
    try:
        hits = np.zeros(64, dtype=ma.HIT_DT)
        hits["qns"] = (np.arange(64, dtype=np.uint64) % np.uint64(8)) << np.uint64(32)
        hits["tn"] = (np.arange(64) + 1) % 8
        ctx.hits_upload(hits, 8)
        b = (C.c_uint32 * 3)(0, 5, 8)
        ma._chk(L.mahip_set_shard_bounds(ctx.h, b, 2), "set_shard_bounds")
        w = C.c_int(-1)
        p = L.mahip_shard_bounds(ctx.h, C.byref(w))
        assert w.value == 2 and [p[0], p[1], p[2]] == [0, 5, 8]
        ctx.hits_upload(hits, 8)  # the same number of reads: until round 4 the table stayed
        p = L.mahip_shard_bounds(ctx.h, C.byref(w))
        assert w.value == 0 and not p, "a table of read ranges survived the next upload"
    finally:
        ctx.close()
