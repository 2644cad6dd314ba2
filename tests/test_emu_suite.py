"""The `-m gpu` parity tests, run WITHOUT a GPU: the product's kernel sources (miniasm_amd/csrc/*.hip, unmodified) are
compiled a second time for the CPU against tests/emu -- a fiber-based stand-in for the HIP runtime that models wave64
shuffles, ballots, DPP controls, barriers and atomics -- and the GPU test modules are run against that build in a
subprocess (`-p emu_plugin` swaps the library path of the ctypes harness).  This is test infrastructure: it proves the
kernels' logic (every stage bit-exact against the oracle and the reference library) on the box that has no GPU; it says
nothing about speed, stream ordering or memory-model behaviour, which only the `-m gpu` run on an MI355X covers.

By default a subset that finishes in about three minutes runs; MA_EMU_FULL=1 runs every GPU test module except the
BASELINE-scale inputs and the RCCL tests (about 25 minutes on 8 cores)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
FULL = os.environ.get("MA_EMU_FULL", "0") == "1"


@pytest.fixture(scope="module")
def emu_built(built):
    r = subprocess.run(["make", "-C", EMU, "-j8", "all", "_build/emu_selftest"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    return True


def run_gpu_tests(args, timeout, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env["PYTHONPATH"] = EMU + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("MINIASM_AMD_LIB", None)
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-p", "emu_plugin", "-x", "-q", "-p", "no:cacheprovider"] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = r.stdout[-6000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
    return tail


def test_emulator_selftest(emu_built):
    """the stand-in itself: shuffles, ballot, every DPP control the kernels use, barriers with early exits, divergent loops"""
    r = subprocess.run([os.path.join(EMU, "_build", "emu_selftest")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout


def test_kernels_stage_parity_on_cpu(emu_built):
    """tests/test_gpu_parity.py: every HIP pass against the oracle and the reference library (incl. the second tiers)"""
    sel = [] if FULL else ["-k", "lognormal or noisy or lowid or deep_groups or sort_random or sub_ or random_hit"]
    run_gpu_tests(["tests/test_gpu_parity.py"] + sel, 3000)


@pytest.mark.parametrize("stride", ["1", "0"])
def test_sort_does_not_depend_on_the_run_stride_hint(stride, emu_built):
    """the hit sort takes RUNS of records when told how a query's own records stand in the array (mahip_set_run_stride; the stage tests default to 2 = ma_hit_read's
    layout).  A hint that does not fit the data -- stride 1 on mirrored records: too few runs; random hit arrays under any stride -- must cost time, never correctness:
    the same stage tests with the other hints (0 = no hint: one key per record)"""
    run_gpu_tests(["tests/test_gpu_parity.py", "-k", "lognormal or noisy or sort_random or random_hit or deep_groups"], 3000, {"MA_TEST_RUN_STRIDE": stride})


def test_kernels_with_reversed_schedule_and_guard_pages(emu_built):
    """the same kernels with lanes, waves and blocks executed in DESCENDING order (code that leans on lock-step execution or on launch order
    without a barrier breaks) and every device allocation ending at a faulting page (an out-of-bounds access crashes)"""
    run_gpu_tests(["tests/test_gpu_parity.py", "-k", "noisy or deep_groups or sort_random", "tests/test_gpu_ingest.py"], 3000,
                  {"EMU_ORDER": "reverse", "EMU_GUARD": "1", "MA_DEV_POOL": "0"})  # (the pool hands out pieces of bigger allocations: no guard page behind them)


def test_kernels_graph_api_on_cpu(emu_built):
    """tests/test_gpu_graph_api.py, test_gpu_graph_fuzz.py: device cleaners and unitigs after every call, through the per-symbol ABI (pipeline graphs,
    hand-made rings and hubs, random graphs with random scripts)"""
    run_gpu_tests(["tests/test_gpu_graph_api.py", "tests/test_gpu_graph_fuzz.py"] + ([] if FULL else ["-k", "not noisy_big"]), 1800)


def test_tie_filter_on_cpu(emu_built):
    """tests/test_gpu_cli.py: push conflicts the reference's arc sort cannot see -- the hit walk is skipped, every dump equals the reference's byte for byte; and the
    realistic inputs (jittered coordinates, lines grouped by target) through both walks"""
    run_gpu_tests(["tests/test_gpu_cli.py", "-k", "out_of_sight or in_sight_only or (tie_rich and jitter and default)"], 3000)


def test_kernels_ingest_on_cpu(emu_built):
    """tests/test_gpu_ingest.py: device PAF parser + dictionary against the host reader and the reference"""
    run_gpu_tests(["tests/test_gpu_ingest.py"], 1800)


@pytest.mark.skipif(not FULL, reason="MA_EMU_FULL=1 runs the CLI and sharded suites on the emulator (about 20 minutes)")
def test_cli_suite_on_cpu(emu_built):
    run_gpu_tests(["tests/test_gpu_cli.py", "-k", "not baseline_scale"], 5000)


def test_sharded_suite_on_cpu(emu_built):
    """tests/test_gpu_sharded.py: `MA_GPUS=N miniasm` -- host/sharded.c, N forked ranks over the shared-memory double of the collectives --
    against the single-rank run and the reference binary (tie-rich input included).  The torch-driven virtual-rank test needs a real device."""
    sel = "not rccl and not virtual_ranks" + ("" if FULL else " and (2-lognormal or 3-noisy or tie_order or hold_only or their_own_records or never_leaves)")
    run_gpu_tests(["tests/test_gpu_sharded.py", "-k", sel], 5000)


@pytest.mark.parametrize("mode", ["options", "text", "ranks"])
def test_randomised_reference_vs_kernels_on_cpu(mode, emu_built):
    """tools/fuzz_emu.py, a fixed slice of it: random generator parameters x random options x every dump format ("options"), damaged PAF text
    ("text"), and the same through `MA_GPUS=2` ("ranks"): the reference binary's bytes against the CPU build of the kernels"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")):
        pytest.skip("oracle/_ref not built")
    n = 150 if FULL else 30
    args = {"options": ["--seed", "101"], "text": ["--seed", "102", "--text"], "ranks": ["--seed", "103", "--ranks", "2"]}[mode]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "--cases", str(n)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
    assert r.returncode == 0 and "0 mismatches" in r.stdout.splitlines()[-1], r.stdout[-3000:]


@pytest.mark.parametrize("mode", ["default", "no_tail_ctx"])
def test_bench_py_runs_on_the_cpu_build(mode, emu_built, tmp_path):
    """bench.py itself -- Workload (file -> device parse -> records by read range), Runner with its worker thread, the profiled steps, the
    text-resident leg, the reference run and the GFA comparison, the JSON line -- executed against the CPU build of the kernels with a numpy-backed
    stand-in for the few torch calls it makes (tests/emu/fake_torch).  Default: the second context + hand-over thread; `no_tail_ctx`: one context.
    Numbers mean nothing here; the control flow, the parity check and the shape of the line do."""
    import json
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")):
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(EMU, "fake_torch") + os.pathsep + env.get("PYTHONPATH", "")
    env["MINIASM_AMD_LIB"] = os.path.join(EMU, "_build", "libminiasm_amd_emu.so")
    env["MA_BENCH_DIR"] = str(tmp_path)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "2500", "--lines", "70000", "--seed", "5", "--steps", "3", "--warmup", "1", "--no-legs"]
    if mode == "no_tail_ctx":
        cmd.append("--no-tail-ctx")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["gfa_identical"] is True and d["parity"]["gfa_md5"] == d["parity"]["ref_md5"]
    assert d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] == 1
    assert d["from_text"] and d["from_text"]["value"] > 0
    assert ("second context" in d["config"]["pipelining"]) == (mode == "default")
    assert d["roofline"]["sort_group"]["alg_bytes_per_step"] == 112.0 * d["config"]["per_gpu_hits"] and "k_hit_sub<gather>" in d["roofline"]["sort_group"]["kernels"]
    assert all(k["alg_GBs"] is None for k in d["kernels"] if k["name"] in ("k_hit_keys", "k_radix_scatter", "k_radix_hist", "k_hit_goff"))


@pytest.mark.parametrize("grid", [0, 16])
def test_bench_py_on_two_ranks_on_the_cpu_build(grid, emu_built, tmp_path):
    """`bench.py --gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment), on the CPU build:
    the control plane is a file-based stand-in for torch.distributed, the collectives go through the shared-memory double
    (MA_BENCH_ONE_GPU_DEBUG, the hook bench.py has for one-GPU boxes), rank 0 runs its tail on a second context.  A one-rank run first leaves
    the reference's GFA in the work directory; the two-rank line must say `gfa_identical: true` against it and report whole-job throughput."""
    import json
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "miniasm_ref")):
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(EMU, "fake_torch") + os.pathsep + env.get("PYTHONPATH", "")
    env["MINIASM_AMD_LIB"] = os.path.join(EMU, "_build", "libminiasm_amd_emu.so")
    env["MA_BENCH_DIR"] = str(tmp_path)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "2500", "--lines", "70000", "--seed", "5", "--steps", "2", "--warmup", "1"]
    if grid:  # a tie-rich input: the ranks (which hold only their own records, and say where they stood in the input) have to restore the reference's order of tied hits
        base += ["--grid", str(grid), "--model", "uniform", "--reads", "3000", "--lines", "80000"]
    r = subprocess.run(base + ["--no-legs", "--no-text"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    one = json.loads(r.stdout.strip().splitlines()[-1])
    assert one["gfa_identical"] is True
    assert (one["tie_groups"] > 0) == bool(grid)
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(20000 + os.getpid() % 20000),
                 MA_FAKE_DIST_DIR=str(tmp_path), MA_BENCH_ONE_GPU_DEBUG="1")
        procs.append(subprocess.Popen(base + ["--gpus", "2"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, err = p.communicate(timeout=1800)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:]
        outs.append(o)
    assert outs[1].strip() == "", "only rank 0 prints"
    d = json.loads(outs[0].strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["gfa_identical"] is True and d["parity"]["gfa_md5"] == one["parity"]["ref_md5"]
    assert d["tie_groups"] == one["tie_groups"] and (not grid or "hit walk" in d["tie_path"])  # tie-rich: both walks ran on the shards too
    assert d["config"]["global_overlaps"] == one["config"]["global_overlaps"] and d["config"]["per_gpu_hits"] < one["config"]["per_gpu_hits"]
    ph = d["phases"]  # where a sharded step spends its time: one entry per phase of host/sharded.c, [max, min] over the ranks
    assert set(ph["phase_ms"]) >= {"sort", "sub#1", "x:sub0", "x:arc blocks", "rank 0: cleanup+symm"} and all(len(v) == 2 and v[0] >= v[1] >= 0 for v in ph["phase_ms"].values())
    assert ph["exchange_bytes_per_rank"]["x:sub0"] > 0 and ph["head_wall_ms_max"] >= ph["head_wall_ms_min"] > 0 and ph["rank0_tail_wall_ms"] > 0
