"""CPU, world_size 2 and 3 over gloo: the sharded-mode orchestration (miniasm_amd/sharded.py: read-range shards,
sub all-gathers, flag max-all-reduces, the arc all-gather, ranged transitive reduction, del-flag all-gather) driven
with the oracle-backed stand-in must give exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, paf, out_path):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import miniasm_amd as ma
    from miniasm_amd.sharded import Comm, run_sharded
    from dist_double import OracleBackend
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    be = OracleBackend(ing.hits, ing.n_seq)
    stats = run_sharded(be, Comm(), opt, ing.n_seq)
    if rank == 0:
        np.savez(out_path, arcs=be.result_arcs_squeezed(), sub=be.subs[0], rdel=be.r_del, **{k: np.int64(v) for k, v in stats.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["lognormal", "noisy"])
def test_sharded_orchestration_matches_single_process(world, case, tmpdir_s):
    sys.path.insert(0, HERE)
    import miniasm_amd as ma
    import refapi as R
    import stages as ST
    extra = [] if case == "lognormal" else ["-L", "uniform", "-d", "0.35", "-x", "0.03"]
    paf = R.pafgen(os.path.join(tmpdir_s, "dist_%s.paf" % case), 1201, 30000, 71, extra)  # 1201 reads: uneven shards
    out = os.path.join(tmpdir_s, "dist_%s_%d.npz" % (case, world))
    mp.spawn(_worker, args=(world, _free_port(), paf, out), nprocs=world, join=True)
    got = np.load(out)
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    one = ST.orc_stages(ing.hits, ing.n_seq, opt)
    assert int(got["n_rem1"]) == one["n_rem1"] and int(got["n_rem2"]) == one["n_rem2"]
    assert int(got["n_seq_new"]) == one["n_seq_new"] and int(got["n_hits"]) == len(one["cont"])
    assert int(got["n_arc"]) == len(one["sg_arcs"]) and int(got["n_red"]) == one["tr_cnt"]["n_red"]
    assert (int(got["n_multi"]), int(got["n_asymm"])) == (one["tr_cnt"]["n_multi"], one["tr_cnt"]["n_asymm"])
    assert got["sub"].tobytes() == one["subm"].tobytes()
    assert (got["rdel"] == 0).tobytes() == (one["map"] >= 0).tobytes()
    assert got["arcs"].tobytes() == one["tr_arcs"].tobytes(), "reduced graph differs from the single-process result"
    ing.close()


def test_single_rank_path_without_process_group():
    """world 1 (no process group): the same code path with every exchange a no-op"""
    sys.path.insert(0, HERE)
    import miniasm_amd as ma
    import refapi as R
    import stages as ST
    from miniasm_amd.sharded import Comm, run_sharded, shard_range
    from dist_double import OracleBackend
    assert shard_range(10, 3, 0) == (4, 0, 4) and shard_range(10, 3, 2) == (4, 8, 10) and shard_range(2, 4, 3) == (1, 2, 2)
    import tempfile
    paf = R.pafgen(os.path.join(tempfile.mkdtemp(), "one.paf"), 800, 20000, 72, [])
    opt = ma.default_opt()
    ing = ma.Ingest(paf, opt)
    be = OracleBackend(ing.hits, ing.n_seq)
    run_sharded(be, Comm(), opt, ing.n_seq)
    one = ST.orc_stages(ing.hits, ing.n_seq, opt)
    assert be.result_arcs_squeezed().tobytes() == one["tr_arcs"].tobytes()
    ing.close()
