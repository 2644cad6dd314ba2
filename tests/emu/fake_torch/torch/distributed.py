"""see torch/__init__.py in this directory.  bench.py uses torch.distributed only as its control plane (rendezvous of the communicator id,
barriers, the maximum over ranks of the measured time); this stand-in does the same through files in a directory named after MASTER_PORT, so
that `bench.py --gpus N` can be executed as N processes on the CPU build of the kernels (collectives: the shared-memory double)."""
import os
import pickle
import time


class ReduceOp:
    MAX = "max"
    SUM = "sum"
    MIN = "min"


_S = {"rank": 0, "world": 1, "dir": None, "seq": 0}


def _wait(path, timeout=600.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("fake torch.distributed: timed out waiting for " + path)
        time.sleep(0.002)


def _exchange(obj):
    """every rank contributes one object; returns the list of all of them, by rank"""
    _S["seq"] += 1
    base = os.path.join(_S["dir"], "x%06d" % _S["seq"])
    tmp = "%s.r%d.tmp" % (base, _S["rank"])
    with open(tmp, "wb") as f:
        pickle.dump(obj, f)
    os.replace(tmp, "%s.r%d" % (base, _S["rank"]))
    out = []
    for r in range(_S["world"]):
        p = "%s.r%d" % (base, r)
        _wait(p)
        with open(p, "rb") as f:
            out.append(pickle.load(f))
    return out


def init_process_group(backend=None, **kw):
    _S["rank"], _S["world"] = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    _S["dir"] = os.path.join(os.environ.get("MA_FAKE_DIST_DIR", "/tmp"), "fake_dist_%s" % os.environ.get("MASTER_PORT", "0"))
    os.makedirs(_S["dir"], exist_ok=True)
    _exchange("hello")


def barrier():
    _exchange(None)


def broadcast_object_list(box, src=0):
    got = _exchange(list(box) if _S["rank"] == src else None)
    box[:] = got[src]


def all_reduce(t, op=ReduceOp.MAX):
    import numpy as np
    vals = _exchange(t.a.copy())
    acc = vals[0].copy()
    for v in vals[1:]:
        acc = np.maximum(acc, v) if op == ReduceOp.MAX else np.minimum(acc, v) if op == ReduceOp.MIN else acc + v
    t.a[...] = acc


def destroy_process_group():
    pass
