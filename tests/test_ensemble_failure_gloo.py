"""A failed run on ONE rank must fail the whole job cleanly: the records (with their status column) are
all-gathered first and every rank raises afterwards -- raising before the collective would leave the
other ranks blocked in the all-gather until the communicator times out.  Two processes,
gloo, world_size 2; plus the world_size-1 process group (the collective must run there too: it is what
the single-GPU bench and the `-m gpu` RCCL test go through)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
from dynesty_amd import backend, ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
bad_rank = int(os.environ["BAD_RANK"])
raise_rank = int(os.environ.get("RAISE_RANK", "-1"))
TOTAL = int(os.environ.get("TOTAL_RUNS", "5"))

class FakeBackend:
    def ns_ensemble(self, prob, runs, nlive, queue_size, first_run=0, want_samples=False, **kw):
        if rank == raise_rank:
            raise MemoryError("libdynhip error -5: hipMalloc failed")  # what _check raises for DH_ERR_NOMEM-like codes
        status = np.zeros(runs, dtype=np.int32)
        if rank == bad_rank:
            status[-1] = -2
        out = dict(logz=-57.5 + 0.01 * (first_run + np.arange(runs)), logzerr=np.full(runs, 0.1),
                   niter=np.full(runs, 10), ncall=np.full(runs, 100), h=np.full(runs, 28.0), status=status)
        if want_samples:
            d = 2
            out.update(dead_logl=np.sort(np.random.default_rng(first_run).random((runs, 10)), axis=1),
                       dead_u=np.random.default_rng(1).random((runs, 10, d)),
                       live_logl=1.0 + np.random.default_rng(2).random((runs, nlive)),
                       live_u=np.random.default_rng(3).random((runs, nlive, d)),
                       dead_id=np.zeros((runs, 10), dtype=np.int64), dead_it=np.ones((runs, 10), dtype=np.int64),
                       dead_nc=np.ones((runs, 10), dtype=np.int64), live_it=np.zeros((runs, nlive), dtype=np.int64))
        return out

    def problem_eval(self, prob, u):
        return np.asarray(u), None

backend.set_backend(FakeBackend())
dist.init_process_group("gloo")
res = {}
try:
    t = ensemble.run_ensemble_device(None, TOTAL, world=world, rank=rank, dist=dist)
    res["device"] = ["ok", t[:, 1].tolist()]
except (RuntimeError, MemoryError) as e:
    res["device"] = ["raised", type(e).__name__ + ": " + str(e)]
if raise_rank < 0:
    t = ensemble.run_ensemble_device(None, TOTAL, world=world, rank=rank, dist=dist, on_failure='nan')
    res["nan"] = [int(np.isnan(t[:, 1]).sum()), t.shape]
try:
    m = ensemble.run_ensemble_merged_sharded(None, TOTAL, world=world, rank=rank, dist=dist, nlive=8, queue_size=4)
    res["merged"] = ["ok", int(m.niter)]
except (RuntimeError, MemoryError) as e:
    res["merged"] = ["raised", type(e).__name__ + ": " + str(e)]
dist.barrier()
with open(os.path.join(%(out)r, "f%%d_of_%%d.json" %% (rank, world)), "w") as f:
    json.dump(res, f)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp, world, bad_rank, raise_rank=-1, total=5):
    import json
    script = os.path.join(str(tmp), "worker.py")
    with open(script, "w") as f:
        f.write(WORKER % dict(root=ROOT, out=str(tmp)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", BAD_RANK=str(bad_rank),
                   RAISE_RANK=str(raise_rank), TOTAL_RUNS=str(total))
        procs.append(subprocess.Popen([sys.executable, script], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0  # a hang in the collective would time out here
    return [json.load(open(os.path.join(str(tmp), f"f{r}_of_{world}.json"))) for r in range(world)]


def test_a_failed_run_on_one_rank_raises_on_every_rank(tmp_path):
    res = _launch(tmp_path, 2, bad_rank=1)
    for r in res:
        assert r["device"][0] == "raised" and "status" in r["device"][1], r
        assert "[4]" in r["device"][1]  # the failed run's GLOBAL id, known to both ranks
        assert r["nan"] == [1, [5, 6]]
        assert r["merged"][0] == "raised", r


def test_no_failure_two_ranks(tmp_path):
    res = _launch(tmp_path, 2, bad_rank=-1)
    assert res[0] == res[1]
    assert res[0]["device"][0] == "ok" and len(res[0]["device"][1]) == 5
    assert res[0]["nan"] == [0, [5, 6]]
    assert res[0]["merged"] == ["ok", 5 * 18]


def test_world_size_one_process_group_runs_the_collectives(tmp_path):
    res = _launch(tmp_path, 1, bad_rank=-1)
    assert res[0]["device"][0] == "ok" and res[0]["merged"] == ["ok", 5 * 18]
    res = _launch(tmp_path, 1, bad_rank=0)
    assert res[0]["device"][0] == "raised" and res[0]["merged"][0] == "raised"


def test_an_exception_on_one_rank_raises_on_every_rank(tmp_path):
    """ns_ensemble RAISING on one rank (argument / memory / HIP errors have no status row) used to
    leave before the collective and block the others; now its runs travel with a sentinel status and every rank
    raises -- the owner its own exception, the others a RuntimeError naming the runs."""
    res = _launch(tmp_path, 2, bad_rank=-1, raise_rank=1)
    assert res[1]["device"][0] == "raised" and res[1]["device"][1].startswith("MemoryError"), res[1]
    assert res[0]["device"][0] == "raised" and "raised on the rank" in res[0]["device"][1], res[0]
    assert "[3, 4]" in res[0]["device"][1]
    assert res[0]["merged"][0] == "raised" and res[1]["merged"][0] == "raised"
    assert res[1]["merged"][1].startswith("MemoryError")


def test_failures_at_world_size_8(tmp_path):
    """the first real 8-GPU run must not be the first time a failing rank meets seven healthy
    ones.  20 runs over 8 ranks (ragged shards: 3, 3, 3, 3, 2, 2, 2, 2): (i) the last run of rank 5 ends with a bad
    status -- every rank raises after the gather and names the run's global id, 'nan' mode marks exactly that run, the
    merged path raises everywhere; (ii) rank 3's backend raises -- its three runs travel with the sentinel status, the
    owner re-raises its own exception, the other seven a RuntimeError naming those runs; nobody hangs."""
    from dynesty_amd import ensemble
    shards = [ensemble.shard_runs(20, 8, r) for r in range(8)]
    assert [len(s) for s in shards] == [3, 3, 3, 3, 2, 2, 2, 2]
    bad = shards[5].stop - 1
    res = _launch(tmp_path, 8, bad_rank=5, total=20)
    for r in res:
        assert r["device"][0] == "raised" and "status" in r["device"][1] and f"[{bad}]" in r["device"][1], r
        assert r["nan"] == [1, [20, 6]]
        assert r["merged"][0] == "raised", r
    res = _launch(tmp_path, 8, bad_rank=-1, raise_rank=3, total=20)
    ids = list(range(shards[3].start, shards[3].stop))
    for rk, r in enumerate(res):
        assert r["device"][0] == "raised" and r["merged"][0] == "raised", (rk, r)
        if rk == 3:
            assert r["device"][1].startswith("MemoryError") and r["merged"][1].startswith("MemoryError")
        else:
            assert "raised on the rank" in r["device"][1] and str(ids) in r["device"][1], (rk, r)
    # and the healthy case at this shape
    res = _launch(tmp_path, 8, bad_rank=-1, total=20)
    assert all(r == res[0] for r in res)
    assert res[0]["device"][0] == "ok" and len(res[0]["device"][1]) == 20 and res[0]["merged"] == ["ok", 20 * 18]
