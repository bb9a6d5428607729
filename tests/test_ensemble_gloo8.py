"""BASELINE C5 is 512 runs over 8 ranks; the driver's 8-GPU run must not be the first time that shape executes
(VERDICT round 3).  Eight processes, gloo, world_size 8: the 512-run sharding, the all-gather of the records and
the RAGGED gather of every run's points (run lengths differ by rank and by run) followed by the merge on every rank --
against the same ensemble run by ONE process.  The backend is a stand-in whose results are a function of the GLOBAL
run id alone (as the device's are: seeds are keyed on it), so the two must agree exactly."""
import json
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
NLIVE, D, CAP = 12, 3, 40

class KeyedBackend:
    """every run's record and points from its global id alone"""
    def ns_ensemble(self, prob, runs, nlive, queue_size, first_run=0, want_samples=False, **kw):
        ids = first_run + np.arange(runs)
        niter = 20 + (ids * 7) %% 17                     # ragged: 20 .. 36 dead points
        out = dict(logz=-57.5 + 0.01 * np.sin(ids), logzerr=0.1 + 0.001 * (ids %% 5), niter=niter,
                   ncall=1000 + 3 * ids, h=28.0 + 0.01 * (ids %% 11), status=np.zeros(runs, dtype=np.int32))
        if want_samples:
            dl = np.zeros((runs, CAP)); du = np.zeros((runs, CAP, D))
            ll = np.zeros((runs, nlive)); lu = np.zeros((runs, nlive, D))
            for i, g in enumerate(ids):
                rng = np.random.default_rng(1000 + int(g))
                dl[i, :niter[i]] = np.sort(rng.random(niter[i])); du[i] = rng.random((CAP, D))
                ll[i] = 1.0 + rng.random(nlive); lu[i] = rng.random((nlive, D))
            out.update(dead_logl=dl, dead_u=du, live_logl=ll, live_u=lu,
                       dead_id=np.tile(np.arange(CAP) %% nlive, (runs, 1)).astype(np.int64),
                       dead_it=np.tile(np.arange(CAP) + 1, (runs, 1)).astype(np.int64),
                       dead_nc=np.ones((runs, CAP), dtype=np.int64), live_it=np.zeros((runs, nlive), dtype=np.int64))
        return out

    def problem_eval(self, prob, u):
        return np.asarray(u), None

backend.set_backend(KeyedBackend())
if world > 1:
    dist.init_process_group("gloo")
d = dist if world > 1 else None
table = ensemble.run_ensemble_device(None, 512, world=world, rank=rank, dist=d, nlive=NLIVE, queue_size=4)
merged = ensemble.run_ensemble_merged_sharded(None, 512, world=world, rank=rank, dist=d, nlive=NLIVE, queue_size=4,
                                              max_iter=CAP)
mean, se, n = ensemble.combine_logz(table)
out = dict(shard=[int(x) for x in (ensemble.shard_runs(512, world, rank).start, ensemble.shard_runs(512, world, rank).stop)],
           table_sum=float(table.sum()), table_ids=table[:, 0].tolist(), mean=mean, se=se, n=n,
           mniter=int(merged.niter), mlogz=float(merged.logz[-1]), mlogzerr=float(merged.logzerr[-1]),
           mncall=int(np.sum(merged.ncall)), mrun_head=merged.samples_run[:40].tolist(),
           mlogl_sum=float(np.sum(merged.logl)), mu_sum=float(np.sum(merged.samples_u)))
with open(os.path.join(%(out)r, "g%%d_of_%%d.json" %% (rank, world)), "w") as f:
    json.dump(out, f)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp, world):
    script = os.path.join(str(tmp), "worker8.py")
    with open(script, "w") as f:
        f.write(WORKER % dict(root=ROOT, out=str(tmp)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, script], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [json.load(open(os.path.join(str(tmp), f"g{r}_of_{world}.json"))) for r in range(world)]


def test_c5_sharding_and_ragged_merge_at_world_size_8(tmp_path):
    one = _launch(tmp_path, 1)[0]
    eight = _launch(tmp_path, 8)
    assert [e["shard"] for e in eight] == [[64 * r, 64 * (r + 1)] for r in range(8)]  # 512 runs: 64 per rank
    for e in eight:
        # every rank holds the same table and the same merged run, and they are the single-process ones
        for key in ("table_ids", "table_sum", "mean", "se", "n", "mniter", "mlogz", "mlogzerr", "mncall", "mrun_head",
                    "mlogl_sum", "mu_sum"):
            assert e[key] == one[key], (key, e[key], one[key])
    assert one["table_ids"] == [float(i) for i in range(512)] and one["n"] == 512
    assert one["mniter"] == sum(20 + (i * 7) % 17 for i in range(512)) + 512 * 12
