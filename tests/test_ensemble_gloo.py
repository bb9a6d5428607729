"""N > 1 path on CPU: two processes, gloo backend, world_size 2.  Each rank runs
its shard of an ensemble (oracle test backend: no GPU here) and the records /
ragged samples are all-gathered.  Checks: sharding covers every run once,
results are independent of the world size (seeds keyed on the global run id),
the gathered table is identical on both ranks."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
from dynesty_amd import backend, ensemble
from oracle_backend import OracleBackend
import inputs
backend.set_backend(OracleBackend())
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo")
prob = inputs.problem("C1")
table, results = ensemble.run_ensemble(
    prob, 5, base_seed=7, world=world, rank=rank, dist=dist if world > 1 else None,
    nlive=60, bound="single", sample="unif", queue_size=8, dlogz=1.0)
mine = ensemble.shard_runs(5, world, rank)
rag = ensemble.gather_ragged([r.samples_u[:3 + rid] for rid, r in zip(mine, results)],
                             world, rank, dist=dist if world > 1 else None)
rows = []
for r in results:
    k = r.niter
    # final live points back in slot order (run_rows numbers them by position)
    sl = np.argsort(r.samples_id[k:])
    rows.append(ensemble.run_rows(r.samples_logl[:k], r.samples_u[:k], r.samples_logl[k:][sl], r.samples_u[k:][sl],
                                  r.samples_id[:k], r.samples_it[:k], r.samples_nc[:k], r.samples_it[k:][sl]))
merged = ensemble.gather_and_merge(rows, 60, world, rank, dist=dist if world > 1 else None)
assert merged.ncall.shape == merged.logl.shape and (merged.samples_id < 60).all() and (merged.samples_id >= 0).all()
# the per-point columns travelled with their points: every (run, id, it) triple is one of the run's own
for rid, r in zip(mine, results):
    sel = merged.samples_run == rid
    got = set(zip(merged.samples_id[sel].tolist(), merged.samples_it[sel].tolist(), merged.ncall[sel].tolist()))
    assert got == set(zip(r.samples_id.tolist(), r.samples_it.tolist(), r.samples_nc.tolist()))
out = dict(table=table.tolist(), nrag=[len(a) for a in rag],
           rag0=rag[0].tolist(), mlogz=float(merged.logz[-1]), mlogzerr=float(merged.logzerr[-1]),
           mniter=int(merged.niter), mncall=int(merged.ncall.sum()), mu0=merged.samples_u[:5].tolist())
with open(os.path.join(%(out)r, "r%%d_of_%%d.json" %% (rank, world)), "w") as f:
    json.dump(out, f)
if world > 1:
    dist.destroy_process_group()
'''


def _launch(tmp, world, port):
    code = WORKER % dict(root=ROOT, out=str(tmp))
    script = os.path.join(str(tmp), "worker.py")
    with open(script, "w") as f:
        f.write(code)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, script], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0


def test_shard_runs_cover():
    from dynesty_amd import ensemble
    for total in (1, 5, 64, 512, 513):
        for world in (1, 2, 3, 8):
            ids = [i for r in range(world)
                   for i in ensemble.shard_runs(total, world, r)]
            assert ids == list(range(total))
            sizes = [len(ensemble.shard_runs(total, world, r))
                     for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_two_process_gather(tmp_path):
    import json
    _launch(tmp_path, 1, 29611)
    _launch(tmp_path, 2, 29612)
    one = json.load(open(tmp_path / "r0_of_1.json"))
    two0 = json.load(open(tmp_path / "r0_of_2.json"))
    two1 = json.load(open(tmp_path / "r1_of_2.json"))
    t1, t20, t21 = (np.array(x["table"]) for x in (one, two0, two1))
    assert t1.shape == (5, 6)
    np.testing.assert_array_equal(t20, t21)       # same table on every rank
    np.testing.assert_array_equal(t1, t20)        # independent of world size
    np.testing.assert_array_equal(t1[:, 0], np.arange(5))
    assert two0["nrag"] == two1["nrag"] == one["nrag"]
    np.testing.assert_array_equal(np.array(one["rag0"]), np.array(two0["rag0"]))
    # sanity of the physics: C1 truth -8.987
    assert abs(t1[:, 1].mean() + 8.987) < 1.0
    # the merged run (ragged gather of every run's points + merge on each rank) is the same
    # on both ranks and for both world sizes, and tighter than a single run
    for k in ("mlogz", "mlogzerr", "mniter", "mncall"):
        assert one[k] == two0[k] == two1[k], k
    np.testing.assert_array_equal(np.array(one["mu0"]), np.array(two1["mu0"]))
    assert one["mniter"] == int(t1[:, 3].sum()) + 5 * 60
    assert abs(one["mlogz"] + 8.987) < 5 * one["mlogzerr"] + 0.2
    assert one["mlogzerr"] < np.mean(t1[:, 2])
