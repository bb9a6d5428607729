#!/usr/bin/env python
"""bench.py -- headline benchmark of the dynesty bounding + proposal hot path
on MI355X (contract: see the round prompt; metric from BASELINE.json).

Workload (config.workload): BASELINE config C2 -- 25-D rho=0.4 correlated
Normal, nlive=2000, bound='multi', sample='rwalk' (walks = 45) -- as `runs`
independent runs per GPU (64 = one GPU's shard of the C5 ensemble of 512 runs
on 8 GPUs).  One *step* = one bound-update interval of every run of the shard
at the queue size the evidence gate allows (K = 512 walkers in flight per run:
ln Z within 0.05 of the reference, tests/test_gpu_logz_gate.py), all inputs
resident in HBM:

  1. MultiEllipsoid.update on each run's live set        (rebuild kernels)
  2. scale_to_logvol(logvol + ln 1.25)                   (enlarge kernel)
  3. ceil(nlive / K) = 4 queue fills: each ONE launch of runs x K walkers x
     `walks` rwalk proposals against the rebuilt, enlarged ellipsoid frame of
     the walker's run, started from live points of that run (in-kernel
     PCG64/ziggurat draws, frame mat-vec, prior transform, Gaussian
     log-likelihood, accept test)

i.e. what the reference does between two bound updates
(update_interval = walks * nlive = 90 000 calls, dynesty.py:213-232) when its
queue holds K proposals.  The same interval flown as ONE launch of nlive
walkers per run (K = nlive: faster per proposal, but a queue that large
biases ln Z -- in the reference exactly as on the device) is reported beside it
(config.interval_at_queue_nlive; it was the headline of rounds 1-2).

Live sets are uniform-in-contour shells: a run's live points are distributed
uniformly inside {logl > loglstar} (here an ellipsoid well inside the prior
box), which is exactly the distribution of the live points of a real run
(SURVEY.md section 8d), not a Gaussian cloud.

value = proposals/s over all GPUs (weak scaling: per-GPU work fixed);
config.rebuilds_per_s is the second half of BASELINE.json's metric.
Tap point: (A) kernel boundary.

`python bench.py --gpus N` (N > 1, no torchrun) launches the N ranks itself;
under torchrun (WORLD_SIZE set) it is one rank of the job.

After the timed region the very entry points that were timed are run once
more from known generator states and compared with the oracle
(config.verified); a mismatch fails the benchmark.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6  # vector fp64 (datasheet)
MAX_ELLS = 8
GATE_QUEUE = 512  # walkers in flight per run for which ln Z passes the gate against the reference (K = nlive does not)
RWALK_SCALE = 0.27  # acceptance ~0.5 on the contour shells: the reference's tuned state (facc = 0.5)
# the REAL dynesty 3.0.0 on C2, one core of the build container (it cannot
# travel to the GPU box: /root/reference does not exist there)
REFERENCE_C2_PROPOSALS_PER_S_1CORE = 44.0e3
REFERENCE_C2_REBUILDS_PER_S_1CORE = 15.0


def c2_problem():
    from dynesty_amd import problems
    return problems.gauss_corr(25, 0.4, 5.0, "C2")


def make_shard(prob, runs, nlive, seed, radius=4.0):
    """`runs` live sets of nlive points, uniform inside the likelihood contour
    v^T Sigma^-1 v < radius^2 (u = 0.5 + v / 10): the distribution of a real run's
    live points at the iteration whose threshold is that contour.  Returns
    (u0 (runs*nlive, d), loglstar)."""
    d = prob.ndim
    rng = np.random.default_rng(seed)
    cov = np.full((d, d), 0.4)
    np.fill_diagonal(cov, 1.0)
    chol = np.linalg.cholesky(cov)
    hw = prob.prior_par[0]
    z = rng.standard_normal((runs * nlive, d))
    z /= np.linalg.norm(z, axis=1)[:, None]
    rad = radius * rng.random(runs * nlive)**(1.0 / d)
    v = (z * rad[:, None]) @ chol.T
    u0 = 0.5 + v / (2 * hw)
    loglstar = float(prob.like_par[0] - 0.5 * radius * radius)
    return u0, loglstar


class Shard:
    """The device-resident state of one rank's benchmark shard and the three
    timed entry points (dh_rebuild_batch_dev, dh_enlarge_batch_dev,
    dh_rwalk_batch_dev).  tests/test_gpu_bench_shape.py drives the same object."""

    def __init__(self, ctx, prob, runs=64, nlive=2000, walks=45, seed=1000,
                 scale=RWALK_SCALE, entropy=(21, 0, 0, 0), queue=GATE_QUEUE):
        self.ctx, self.prob = ctx, prob
        self.runs, self.nlive, self.walks, self.scale = runs, nlive, walks, scale
        self.d = d = prob.ndim
        self.k = k = runs * nlive
        self.kq = kq = min(int(queue), nlive)   # walkers in flight per run
        self.nq = nq = -(-nlive // kq)          # queue fills per bound-update interval
        self.u0, self.loglstar = make_shard(prob, runs, nlive, seed)
        self.log_enlarge = math.log(1.25)
        self.entropy = list(entropy)
        me = self.me = MAX_ELLS
        self.d_u0 = ctx.to_device(self.u0)  # live sets == walker start points
        self.idx = (np.arange(k, dtype=np.int32) // nlive) * me
        self.d_idx = ctx.to_device(self.idx)
        # queue form: fill j starts walker i of run r from live point (j * kq + i) % nlive of run r
        # (every start point is a live point of the walker's own run, as in Sampler.propose_live)
        pick = (np.arange(nq)[:, None] * kq + np.arange(kq)[None, :]) % nlive          # (nq, kq)
        self.u0q = np.ascontiguousarray(
            self.u0.reshape(runs, nlive, d)[:, pick].transpose(1, 0, 2, 3).reshape(nq, runs * kq, d))
        self.d_u0q = ctx.to_device(self.u0q)
        self.idxq = (np.arange(runs * kq, dtype=np.int32) // kq) * me
        self.d_idxq = ctx.to_device(self.idxq)
        self.states0 = ctx.seed_children(self.entropy, 0, k)
        self.d_rng = ctx.to_device(self.states0)
        self.d_rng2 = ctx.malloc(k * 32)
        self.d_u, self.d_v = ctx.malloc(k * d * 8), ctx.malloc(k * d * 8)
        self.d_logl = ctx.malloc(k * 8)
        self.d_na, self.d_nr = ctx.malloc(k * 4), ctx.malloc(k * 4)
        self.d_nells, self.d_status = ctx.malloc(runs * 4), ctx.malloc(runs * 4)
        self.d_ctrs = ctx.malloc(runs * me * d * 8)
        self.d_covs = ctx.malloc(runs * me * d * d * 8)
        self.d_ams = ctx.malloc(runs * me * d * d * 8)
        self.d_axes = ctx.malloc(runs * me * d * d * 8)
        self.d_axl = ctx.malloc(runs * me * d * 8)
        self.d_lv = ctx.malloc(runs * me * 8)
        self.ph = ctx.problem(prob)

    # ---- the timed entry points ------------------------------------------
    def rebuild(self, enlarge=True):
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.handle
        c._check(lib.dh_rebuild_batch_dev(h, self.runs, self.d_u0, self.nlive, self.d, 0,
                                          self.me, self.d_nells, self.d_status, self.d_ctrs,
                                          self.d_covs, self.d_ams, self.d_axes, self.d_axl,
                                          self.d_lv, None, None))
        if enlarge:
            c._check(lib.dh_enlarge_batch_dev(h, self.runs, self.me, self.d_nells, self.d,
                                              self.d_covs, self.d_ams, self.d_axes, self.d_axl,
                                              self.d_lv, self.log_enlarge))

    def walk(self, i=0, first=0, count=None):
        """One rwalk launch over walkers [first, first + count) of the shard."""
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.handle
        a, b = (self.d_rng, self.d_rng2) if i % 2 == 0 else (self.d_rng2, self.d_rng)
        count = self.k if count is None else count
        d = self.d

        def off(p, stride):
            return p + first * stride
        c._check(lib.dh_rwalk_batch_dev(h, self.ph, count, d, d, off(self.d_u0, d * 8),
                                        self.d_axes, self.runs * self.me,
                                        off(self.d_idx, 4), self.scale, self.loglstar,
                                        self.walks, None, off(a, 32), off(self.d_u, d * 8),
                                        off(self.d_v, d * 8), off(self.d_logl, 8),
                                        off(self.d_na, 4), off(self.d_nr, 4), off(b, 32)))

    def walk_philox(self, i=0):
        """The same launch in the throughput RNG mode (hiprand Philox4x32-10; no generator states in
        HBM): seed = the shard's entropy word, walker key = its index, offset advances per launch."""
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.handle
        d = self.d
        c._check(lib.dh_rwalk_batch_philox_dev(h, self.ph, self.k, d, d, self.d_u0, self.d_axes,
                                               self.runs * self.me, self.d_idx, self.scale,
                                               self.loglstar, self.walks, None, int(self.entropy[0]), 0,
                                               int(i) * 4096, self.d_u, self.d_v, self.d_logl,
                                               self.d_na, self.d_nr))

    def walk_q(self, i=0, j=0):
        """Queue fill j of step i: one rwalk launch of runs x kq walkers (walker (r, w) starts from a
        live point of run r and walks in run r's frame)."""
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.handle
        a, b = (self.d_rng, self.d_rng2) if (i * self.nq + j) % 2 == 0 else (self.d_rng2, self.d_rng)
        d, n = self.d, self.runs * self.kq
        c._check(lib.dh_rwalk_batch_dev(h, self.ph, n, d, d, self.d_u0q + j * n * d * 8, self.d_axes,
                                        self.runs * self.me, self.d_idxq, self.scale, self.loglstar,
                                        self.walks, None, a, self.d_u, self.d_v, self.d_logl,
                                        self.d_na, self.d_nr, b))

    def walk_q_philox(self, i=0, j=0):
        """Queue fill j of step i in the throughput RNG mode."""
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.handle
        d, n = self.d, self.runs * self.kq
        c._check(lib.dh_rwalk_batch_philox_dev(h, self.ph, n, d, d, self.d_u0q + j * n * d * 8, self.d_axes,
                                               self.runs * self.me, self.d_idxq, self.scale,
                                               self.loglstar, self.walks, None, int(self.entropy[0]), 0,
                                               int(i * self.nq + j) * 4096, self.d_u, self.d_v, self.d_logl,
                                               self.d_na, self.d_nr))

    def step(self, i=0, rebuild=True):
        """One bound-update interval at the gate queue size: rebuild + enlarge + nq queue fills."""
        if rebuild:
            self.rebuild()
        for j in range(self.nq):
            self.walk_q(i, j)

    def step_full(self, i=0, rebuild=True):
        """The same interval as ONE launch of nlive walkers per run (queue size = nlive)."""
        if rebuild:
            self.rebuild()
        self.walk(i)

    # ---- results -------------------------------------------------------------
    def fetch_bound(self):
        c, r, me, d = self.ctx, self.runs, self.me, self.d
        return dict(nells=c.from_device(self.d_nells, (r,), np.int32),
                    status=c.from_device(self.d_status, (r,), np.int32),
                    ctrs=c.from_device(self.d_ctrs, (r, me, d), np.float64),
                    covs=c.from_device(self.d_covs, (r, me, d, d), np.float64),
                    ams=c.from_device(self.d_ams, (r, me, d, d), np.float64),
                    axes=c.from_device(self.d_axes, (r, me, d, d), np.float64),
                    axlens=c.from_device(self.d_axl, (r, me, d), np.float64),
                    logvols=c.from_device(self.d_lv, (r, me), np.float64))

    def fetch_walk(self):
        c, k, d = self.ctx, self.k, self.d
        return dict(u=c.from_device(self.d_u, (k, d), np.float64),
                    v=c.from_device(self.d_v, (k, d), np.float64),
                    logl=c.from_device(self.d_logl, (k,), np.float64),
                    accept=c.from_device(self.d_na, (k,), np.int32),
                    reject=c.from_device(self.d_nr, (k,), np.int32))

    def reset_rng(self):
        c = self.ctx
        c._check(c.lib.dh_memcpy_h2d(c.handle, self.d_rng, self.states0.ctypes.data,
                                     self.states0.nbytes))

    # ---- the check of what was timed (the oracle is the checker only) -------------
    def verify(self, check_runs=None, walkers_per_run=64, rtol=1e-9):
        """Run the timed entry points once from the initial generator states and hold
        `check_runs` to the oracle: the rebuilt + enlarged MultiEllipsoid against
        oracle.multi_update + scale_multi_to_logvol (nells exact, centres 1e-13, cov /
        logvol / axis lengths `rtol`), and the first `walkers_per_run` walkers of each
        of those runs against oracle.rwalk on the same child streams (accept / reject
        counts exact, u within 1e-12, logl 1e-11 relative).  Raises AssertionError."""
        from oracle import bounding_ref as B
        from oracle import proposals_ref as P
        runs, nlive, d = self.runs, self.nlive, self.d
        if check_runs is None:  # every eighth run and the last one (round 4; three runs before)
            check_runs = sorted(set(range(0, runs, 8)) | {runs // 2 - 1 if runs > 2 else 0, runs - 1})
        self.reset_rng()
        self.rebuild(enlarge=False)
        self.ctx.sync()
        plain = self.fetch_bound()
        self.rebuild(enlarge=True)
        self.walk_q(0, 0)        # the timed launch shape: runs x kq walkers, queue fill 0
        self.ctx.sync()
        bnd, wkq = self.fetch_bound(), self.fetch_walk()
        self.reset_rng()
        self.walk(0)             # the K = nlive launch (config.interval_at_queue_nlive)
        self.ctx.sync()
        wk = self.fetch_walk()
        assert np.all(bnd["status"] == 0), bnd["status"]
        assert np.all(wk["accept"] + wk["reject"] == self.walks)
        kq = self.kq
        assert np.all((wkq["accept"] + wkq["reject"])[:runs * kq] == self.walks)
        # EVERY walker of the timed launch: inside the cube, v = prior(u), ln L = L(v), and above the threshold wherever
        # a step was accepted (the oracle's vectorised prior and likelihood: cheap for all runs x kq walkers)
        nq = runs * kq
        uq = wkq["u"][:nq]
        assert np.all((uq > 0.0) & (uq < 1.0))
        v_all = self.prob.prior_transform_many(uq)
        np.testing.assert_allclose(wkq["v"][:nq], v_all, rtol=0, atol=2e-11)
        np.testing.assert_allclose(wkq["logl"][:nq], self.prob.loglikelihood_many(v_all), rtol=1e-11, atol=1e-11)
        moved = wkq["accept"][:nq] > 0
        assert np.all(wkq["logl"][:nq][moved] > self.loglstar) and moved.mean() > 0.9
        kids = np.random.SeedSequence(self.entropy).spawn(self.k)
        nwalk = 0
        for r in check_runs:
            pts = self.u0[r * nlive:(r + 1) * nlive]
            m0 = B.multi_update(pts)
            # scale_multi_to_logvol works in place on the Ell records: scale a copy
            m1 = B.scale_multi_to_logvol(B.stack_ells([e.copy() for e in m0.ells]),
                                         m0.logvol + self.log_enlarge)
            for got, ref in ((plain, m0), (bnd, m1)):
                assert int(got["nells"][r]) == ref.nells, (r, got["nells"][r], ref.nells)
                order = [int(np.argmin(np.linalg.norm(got["ctrs"][r, :ref.nells] - e.ctr, axis=1)))
                         for e in ref.ells]
                assert sorted(order) == list(range(ref.nells))
                for e, j in zip(ref.ells, order):
                    np.testing.assert_allclose(got["ctrs"][r, j], e.ctr, rtol=0, atol=1e-13)
                    np.testing.assert_allclose(got["covs"][r, j], e.cov, rtol=rtol,
                                               atol=rtol * np.abs(e.cov).max())
                    np.testing.assert_allclose(got["ams"][r, j], e.am, rtol=0,
                                               atol=1e-8 * np.abs(e.am).max())
                    np.testing.assert_allclose(got["logvols"][r, j], e.logvol, rtol=0, atol=1e-9)
                    np.testing.assert_allclose(np.sort(got["axlens"][r, j]), np.sort(e.axlens),
                                               rtol=rtol)
                    ax = got["axes"][r, j]
                    np.testing.assert_allclose(ax @ ax.T, e.cov, rtol=0,
                                               atol=1e-10 * np.abs(e.cov).max())
            # walkers of run r use frame r * MAX_ELLS (ellipsoid 0 of the run); the oracle walks
            # in the device's frame (its columns equal the oracle's up to LAPACK's arbitrary signs,
            # checked through axes @ axes.T above)
            frame = bnd["axes"][r, 0]
            # (launch, walker index in the launch, its start point, its child stream)
            cases = [(wkq, r * kq + i, self.u0q[0, r * kq + i], kids[r * kq + i])
                     for i in range(min(walkers_per_run, kq))]
            cases += [(wk, r * nlive + i, self.u0[r * nlive + i], kids[r * nlive + i])
                      for i in range(min(max(walkers_per_run // 4, 1), nlive))]
            for got, w, start, kid in cases:
                rng = np.random.Generator(np.random.PCG64(kid))
                ref = P.rwalk(start.copy(), self.loglstar, frame, self.scale,
                              self.prob.prior_transform, self.prob.loglikelihood, rng, self.walks)
                assert ref["accept"] == got["accept"][w], (w, ref["accept"], got["accept"][w])
                assert ref["reject"] == got["reject"][w], (w, ref["reject"], got["reject"][w])
                np.testing.assert_allclose(got["u"][w], ref["u"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(got["logl"][w], ref["logl"], rtol=1e-11)
                nwalk += 1
        return {"checker": "oracle/ (NumPy restatement pinned to the reference's golden vectors)",
                "runs_checked": [int(r) for r in check_runs],
                "ellipsoids": "nells exact; ctr 1e-13; cov/axlens/logvol 1e-9 (before and after "
                              "the 1.25 enlargement)",
                "walkers_checked": nwalk,
                "all_walkers": f"all {nq} walkers of the timed launch: inside the cube, v = prior(u) 2e-11, "
                               "ln L = L(v) 1e-11, above the threshold wherever a step was accepted",
                "walkers": "accept/reject counts exact; u 1e-12 abs; logl 1e-11 rel; per checked run "
                           f"{min(walkers_per_run, kq)} walkers of the timed launch shape (runs x {kq}) and "
                           f"{min(max(walkers_per_run // 4, 1), nlive)} of the K = nlive launch",
                "ok": True}


# --------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without torchrun
# --------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, extra_env=None):
    """Start n copies of this script, one per GPU (RANK = LOCAL_RANK = 0..n-1,
    rendezvous on 127.0.0.1), wait for all of them; rank 0's stdout (the JSON line) is
    passed through.  Returns the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n),
                    "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if extra_env:
            env.update(extra_env)
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=env, stdout=out))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _gather_ranks(dist, torch, rank, world, device):
    """The ensemble's exchange primitive: an all_gather across the ranks; returns the
    number of distinct ranks that answered."""
    t = torch.tensor([float(rank)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return len({int(x.item()) for x in out})


def launch_selftest(args):
    """No device work: rendezvous, barrier / max-over-ranks timing and the record gather over
    gloo.  Exercises the multi-rank plumbing of this file on a CPU box (tests/test_bench_launch.py);
    the line it prints says so."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group("gloo")
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = _gather_ranks(dist, torch, rank, world, "cpu")
    if rank == 0:
        print(json.dumps({"selftest": "launch plumbing only, NO device work", "n_gpus": world,
                          "rccl_ranks": n, "backend": "gloo", "seconds": float(t.item()),
                          "value": None}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=350)  # > 1 s of timed region at 3.1 ms per step
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--runs", type=int, default=64,
                    help="independent C2 runs per GPU (C5 shard = 64)")
    ap.add_argument("--nlive", type=int, default=2000)
    ap.add_argument("--walks", type=int, default=45)
    ap.add_argument("--queue", type=int, default=GATE_QUEUE,
                    help="walkers in flight per run (the queue size of the evidence gate)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--preroll", type=int, default=80,
                    help="untimed steps before the warm-up (GPU clock ramp)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-rebuild", action="store_true",
                    help="time the proposal kernel alone (diagnostic)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end device-loop legs (tap C)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the timed entry points (diagnostic)")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: only the timed launch shape (no CPU / end-to-end / oracle-check / "
                         "queue-nlive / Philox legs), so that per-kernel averages are those of the headline step")
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.lean:
        args.no_cpu = args.no_e2e = args.no_verify = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launch_selftest:
        return launch_selftest(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # stdout carries ONE line, the JSON record.  RCCL prints a version banner through C stdio on fd 1
    # (flushed at exit, i.e. after the record), so fd 1 is pointed at stderr for the life of the process
    # and the record is written to the saved descriptor.
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)
    prob = c2_problem()
    # CPU baseline first, while this process holds no HIP / torch state: the all-cores leg forks
    # its workers (256 spawned interpreters took 30 s to start on the GPU box's host)
    cpu = None
    if not args.no_cpu and world == 1 and rank == 0:
        u0c, loglstar_c = make_shard(prob, args.runs, args.nlive, 1000 + rank)
        cpu = cpu_baseline(prob, u0c, args.nlive, RWALK_SCALE, loglstar_c, args.walks, args.cpu_seconds)
        del u0c
    import torch
    # LOCAL_RANK indexes the devices this process can SEE: with HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES pre-set by a
    # scheduler the visible list is renumbered from 0 for torch and for libdynhip alike (same runtime rule), so the
    # only thing that can go wrong is a rank without a device of its own -- said here, not as an ordinal error later
    nvis = torch.cuda.device_count()
    if local_rank >= nvis:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {nvis} device(s) are visible "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                         f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # One process group for every world size, 1 included: the record exchange below is then always an
    # RCCL collective that ran on this device (`rccl_ranks` is an observed count, never a constant).
    import torch.distributed as dist
    if world == 1 and "MASTER_ADDR" not in os.environ:
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm

    from dynesty_amd import _lib
    ctx = _lib.Context(local_rank)
    d = prob.ndim
    runs, nlive = args.runs, args.nlive
    sh = Shard(ctx, prob, runs, nlive, args.walks, seed=1000 + rank, entropy=(21, rank, 0, 0), queue=args.queue)
    k, kq, nq = sh.k, sh.kq, sh.nq

    ev = [ctx.event() for _ in range(4)]

    def barrier():
        # own work first (the kernels run on the context's stream, which torch does not
        # see), then the cross-rank barrier, then a device-wide synchronize
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    sh.rebuild()  # frames must exist even with --no-rebuild
    # untimed pre-roll before the W warm-up steps: the first ~0.2 s after the context is created run at
    # ramping clocks (20 timed steps right after 5 warm-up steps measured 4.5 ms, the same steps 3.4 ms
    # once the device is busy); not part of W, not part of the timed region
    for i in range(args.preroll):
        sh.step(i, rebuild=not args.no_rebuild)
    for i in range(args.warmup):
        sh.step(i, rebuild=not args.no_rebuild)
    barrier()
    t0 = time.perf_counter()
    ctx.record(ev[3])
    for i in range(args.steps):
        sh.step(i, rebuild=not args.no_rebuild)
    ctx.record(ev[0])
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ctx.elapsed_ms(ev[3], ev[0]) / args.steps
    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    # per-kernel durations (HIP events on the launch stream), outside the
    # timed region so the event records do not perturb it
    nrep = 5
    t_rb = t_wk = 0.0
    for i in range(nrep):
        ctx.record(ev[0])
        if not args.no_rebuild:
            sh.rebuild()
        ctx.record(ev[1])
        for j in range(nq):
            sh.walk_q(i, j)
        ctx.record(ev[2])
        ctx.sync()
        t_rb += ctx.elapsed_ms(ev[0], ev[1])
        t_wk += ctx.elapsed_ms(ev[1], ev[2]) / nq   # one launch of runs x kq walkers
    t_rb /= nrep
    t_wk /= nrep

    wkr = sh.fetch_walk()
    bnd = sh.fetch_bound()
    nacc, nrej = wkr["accept"][:runs * kq], wkr["reject"][:runs * kq]
    status, nells = bnd["status"], bnd["nells"]
    assert np.all(nacc + nrej == args.walks)
    assert np.all(status == 0), status
    props_per_launch = runs * kq * args.walks
    props_per_step_rank = props_per_launch * nq
    value = world * props_per_step_rank * args.steps / wall

    # The same interval as ONE launch of nlive walkers per run (queue size = nlive: the headline of rounds
    # 1-2), and both shapes in the throughput RNG mode (hiprand Philox) -- outside the timed region
    full = ph = None
    if not args.lean:
        def timed(fn_rb, fn_wk, nwk):
            trb = twk = 0.0
            for i in range(nrep + 1):
                ctx.record(ev[0])
                if not args.no_rebuild:
                    fn_rb()
                ctx.record(ev[1])
                for j in range(nwk):
                    fn_wk(i, j)
                ctx.record(ev[2])
                ctx.sync()
                if i:  # first pass: warm-up
                    trb += ctx.elapsed_ms(ev[0], ev[1]) / nrep
                    twk += ctx.elapsed_ms(ev[1], ev[2]) / nrep
            return trb, twk
        trb, twk = timed(sh.rebuild, lambda i, j: sh.walk(i), 1)
        wk_full = sh.fetch_walk()
        assert np.all(wk_full["accept"] + wk_full["reject"] == args.walks)
        full = {"what": f"the same bound-update interval as ONE launch of {runs} x {nlive} walkers after one "
                        f"rebuild (queue size = nlive; ln Z at this queue size is biased by +0.2 in the "
                        f"reference and on the device alike, tests/test_gpu_logz_gate.py)",
                "ms": trb + twk, "rwalk_kernel_ms": twk,
                "kernel": "itemgen_kernel + rwalkq_kernel<7,PREC_AFFINE,ITEMS> (the form no longer depends on the launch size)",
                "proposals_per_s": world * k * args.walks / ((trb + twk) * 1e-3),
                "proposals_per_s_rwalk_kernel_only": world * k * args.walks / (twk * 1e-3),
                "hbm_frac_algorithmic": k * args.walks * 8 * (2 * d + 1) / (twk * 1e-3) / 1e9 / HBM_PEAK_GBS}
        trb, twk = timed(sh.rebuild, lambda i, j: sh.walk_q_philox(i, j), nq)
        wq = sh.fetch_walk()
        assert np.all((wq["accept"] + wq["reject"])[:runs * kq] == args.walks)
        trb2, twk2 = timed(sh.rebuild, lambda i, j: sh.walk_philox(i), 1)
        ph = {"rng": "hiprand Philox4x32-10, fp32 Box-Muller normals (dh_rwalk_batch_philox_dev); "
                     "not stream-compatible, validated statistically (tests/test_gpu_philox.py)",
              "rwalk_kernel_ms": twk / nq, "step_ms": trb + twk,
              "proposals_per_s_rwalk_kernel_only": world * props_per_launch / (twk / nq * 1e-3),
              "proposals_per_s_step": world * props_per_step_rank / ((trb + twk) * 1e-3),
              "accept_frac": float(wq["accept"][:runs * kq].sum() / props_per_launch),
              "at_queue_nlive": {"rwalk_kernel_ms": twk2, "step_ms": trb2 + twk2,
                                 "proposals_per_s_step": world * k * args.walks / ((trb2 + twk2) * 1e-3)}}

    # saturation: the same step with 32 / 64 / 128 / 256 runs per GPU -- untimed leg
    sweep = None
    if not args.lean and not args.no_rebuild and runs == 64:
        sweep = []
        for rr in (32, 64, 128, 256):
            shs = sh if rr == runs else Shard(ctx, prob, rr, nlive, args.walks, seed=1000 + rank, entropy=(21, rank, 0, 0),
                                              queue=args.queue)
            for i in range(6):
                shs.step(i)
            ctx.record(ev[0])
            for i in range(12):
                shs.step(i)
            ctx.record(ev[1])
            ctx.sync()
            ms = ctx.elapsed_ms(ev[0], ev[1]) / 12
            sweep.append({"runs_per_gpu": rr, "ms_per_step": ms,
                          "proposals_per_s": rr * shs.kq * shs.nq * args.walks / (ms * 1e-3),
                          "rebuilds_per_s": rr / (ms * 1e-3)})
            if shs is not sh:
                del shs

    verified = None
    if not args.no_verify and not args.no_rebuild:
        verified = sh.verify()  # raises on any mismatch

    # ensemble exchange step (C5): one record per run gathered over RCCL -- at every world size
    rec = torch.tensor(nacc.reshape(runs, -1).mean(1), device="cuda")
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    rccl_ranks = _gather_ranks(dist, torch, rank, world, dev)

    # ---- tap C (outside the timed region): the same shard run END TO END by the
    # device-resident nested-sampling loop, every run to dlogz = 0.01
    e2e = None
    if not args.no_e2e:
        from dynesty_amd import ensemble

        def e2e_leg(rebuild_sync, rng="pcg64", forced_exact=True):
            t0 = time.perf_counter()
            table = ensemble.run_ensemble_device(
                prob, runs * world, base_seed=21, world=world, rank=rank,
                dist=dist, device=dev, nlive=nlive, queue_size=GATE_QUEUE, walks=args.walks,
                rebuild_sync=rebuild_sync, rng=rng, forced_exact=forced_exact)
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            lz = table[:, 1]
            return {"runs": int(len(table)), "seconds": dt,
                    "likelihood_calls_per_s": float(table[:, 4].sum() / dt),
                    "ns_iterations_per_s": float(table[:, 3].sum() / dt),
                    "logz_mean": float(lz.mean()),
                    "logz_se": float(lz.std(ddof=1) / math.sqrt(len(lz)))}

        # reference bound-update schedule per run (results independent of the sharding) ...
        e2e = {"tap_point": "C (device-resident NS loop, dh_ns_ensemble)", "queue_size": GATE_QUEUE,
               "protocol": "the reference's: forced bound update inside the fill that finds the start point outside "
                           "(sampler.py:484-489), regular bound built before the newest point enters (:771-772, "
                           "1176-1185) -- forced_exact, the default since round 5"}
        e2e_leg(False)  # untimed: the first call allocates the state arrays and loads the loop's code objects (+7 %)
        e2e.update(e2e_leg(False))
        e2e.update(reference_logz_gate())
        e2e.update({"logz_truth": -57.5646,
                    "gather": f"RCCL all_gather of the per-run records (7 doubles each) over {world} rank(s)"})
        # ... and with the ensemble's rebuilds synchronised (early, never late)
        e2e["rebuild_sync"] = e2e_leg(True)
        # ... in the late form of the forced update (round 4's default: 9-20 % fewer bound updates than the reference)
        e2e["forced_late"] = e2e_leg(False, forced_exact=False)
        # ... and with the proposals drawn from hiprand Philox streams (throughput RNG mode)
        e2e["throughput_rng"] = e2e_leg(False, rng="philox")
        # BASELINE configs C1 (3-D Normal, single / unif + bootstrap 5), C3 (eggbox 2-D, multi / rslice, nlive 5000) and C4 (200-D iid Normal, Normal prior,
        # single / rslice, nlive 4000) through the same loop, at N = 1 only (no rank waits for another at
        # the end of a scaling run): every run to dlogz = 0.01, the real reference's ensembles beside them
        if rank == 0 and world == 1:
            for name, leg in (("config_C1", c1_leg), ("config_C3", c3_leg), ("config_C4", c4_leg)):
                try:
                    e2e[name] = leg(ctx)
                except Exception as exc:  # a side leg must not take the headline down with it
                    e2e[name] = {"error": repr(exc)}

    if rank == 0:
        alg_bytes = 8 * (2 * d + 1)  # read u, write u', write logl
        flops = 2 * d * d + 8 * d + (d * d + 3 * d)  # frame mat-vec + sym. quad form
        achieved = props_per_launch * alg_bytes / (t_wk * 1e-3) / 1e9
        traffic, traffic_src, traffic_rb, issue = None, None, None, None
        pmc = _profile_path("pmc_traffic.json")
        if os.path.exists(pmc) and runs == 64 and nlive == 2000 and args.walks == 45 and kq == GATE_QUEUE:
            # PMC counters cannot be read from inside this process; the values are
            # the committed rocprofv3 measurement of this same launch shape
            # (tools/pmc_traffic.py: 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)
            with open(pmc) as f:
                pj = json.load(f)
            traffic = pj.get("rwalk_launch_traffic_bytes")  # generator pass + walk kernel
            traffic_rb = pj.get("rebuild_pipeline_bytes_per_launch_sequence")
            traffic_src = (os.path.relpath(pmc, ROOT) + " (2*FETCH_SIZE + WRITE_SIZE, "
                           "separate --pmc passes; walk launch = itemgen_kernel + rwalkq_kernel)")
            pi = _profile_path("pmc_issue.json")
            if os.path.exists(pi):
                # SQ issue / stall counters of the same launch shape (tools/pmc_issue.py, three --pmc passes)
                with open(pi) as f:
                    wl = json.load(f)["workloads"].get("bench", {})
                issue = {k: {f: e[f] for f in ("valu_active_frac", "issue_active_frac", "wait_waitcnt_frac",
                                               "wait_issue_stall_frac", "cycles_per_instruction_per_wavefront") if f in e}
                         for k, e in wl.items() if k.startswith(("rwalkq_kernel", "itemgen_kernel", "k_ell", "k_split",
                                                                 "k_root_parts"))}
        line = {
            "metric": "proposals/sec + ellipsoid-rebuilds/sec, 25-D corr-Normal "
                      "nlive=2000 (multi/rwalk)",
            "value": value,
            "unit": "proposals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": args.preroll,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "config": {
                "workload": f"C2 x {runs} independent runs per GPU (C5 shard); "
                            f"per step and run: 1 MultiEllipsoid rebuild + enlarge 1.25 + "
                            f"{nq} queue fills of K={kq} walkers x {args.walks} rwalk steps "
                            f"(= one bound-update interval at the queue size that passes the "
                            f"ln Z gate); live sets uniform inside the likelihood contour",
                "ndim": d, "nlive": nlive, "walks": args.walks, "queue_size": kq,
                "queue_fills_per_step": nq,
                "runs_per_gpu": runs, "tap_point": "A (kernel boundary)",
                "rebuild_in_step": not args.no_rebuild,
                "rebuilds_per_s": (0.0 if args.no_rebuild else
                                   world * runs * args.steps / wall),
                "rebuild_kernel_ms": t_rb, "rwalk_kernel_ms": t_wk,
                "rwalk_launches_per_step": nq,
                "device_ms_per_step": dev_ms,
                "proposals_per_s_rwalk_kernel_only":
                    world * props_per_launch / (t_wk * 1e-3),
                "rebuilds_per_s_rebuild_kernel_only":
                    world * runs / (t_rb * 1e-3) if t_rb > 0 else None,
                "nells_per_run": float(nells.mean()),
                "accept_frac": float(nacc.sum() / props_per_launch),
                "rng": "PCG64 + ziggurat, stream-identical to numpy.random.Generator (parity mode)",
                "runs_per_gpu_sweep": sweep,
                "tap_B": tap_b_record(),
                "verified": verified,
            },
            "roofline": {
                "bound": "hbm", "kernel": "one walk launch = itemgen_kernel (the walkers' PCG64 item streams, one wavefront "
                                          "per walker) + rwalkq_kernel<7,PREC_AFFINE,ITEMS> (four lanes per walker)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": props_per_launch * alg_bytes,
                "kernel_ms": t_wk, "launches_per_step": nq,
                "note": "algorithmic bytes = 408 B/proposal (SURVEY 8d); the walker stays in registers for all 45 "
                        "steps and the generator pass streams 208 B per proposal through HBM / L2, so the measured "
                        "traffic is now of the order of the algorithmic bytes; the walk kernel itself sits at ~85 % "
                        "of the fp64 matrix-core rate on its padded tiles (DESIGN.md 3.1b), the generator pass is "
                        "issue-bound (valu_active_frac below)",
                "valu_active_frac": (None if not issue else
                                     {k: v.get("valu_active_frac") for k, v in issue.items()
                                      if k.startswith(("rwalkq_kernel", "itemgen_kernel"))}),
                "issue_counters": issue,
                "issue_counters_source": os.path.relpath(_profile_path("pmc_issue.json"), ROOT) + " (rocprofv3 --pmc, SQ block; "
                                         "fractions of SQ_WAVE_CYCLES)" if issue else None,
                "fp64_valu": {
                    "achieved": props_per_launch * flops / (t_wk * 1e-3) / 1e12,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s"},
            },
        }
        if full is not None:
            line["config"]["interval_at_queue_nlive"] = full
        if ph is not None:
            line["config"]["throughput_rng_mode"] = ph
        if not args.no_rebuild and t_rb > 0:
            # the rebuild pipeline (k_root / k_split / k_ell / k_finish) takes most of the
            # step; SURVEY 8d prices it at 8*N*D*P bytes with P = 62 dependency-ordered
            # passes over the live set for C2 (std + per level: 10 k-means + mean/cov +
            # Mahalanobis max, + coverage)
            rb_bytes = 8.0 * nlive * d * 62 * runs
            rb_gbs = rb_bytes / (t_rb * 1e-3) / 1e9
            if e2e and isinstance(e2e.get("config_C4"), dict) and "roofline" in e2e["config_C4"]:
                line["roofline_c4"] = e2e["config_C4"]["roofline"]
            line["roofline_rebuild"] = {
                "bound": "hbm", "kernel": "k_root + levels x (k_split, k_ell) + k_finish",
                "achieved": rb_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rb_gbs / HBM_PEAK_GBS, "traffic": traffic_rb,
                "traffic_unit": "bytes per launch sequence (64 runs)",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": rb_bytes, "kernel_ms": t_rb,
                "note": "a tree of ~55 nodes per run built level by level (critical path: "
                        "6 levels x [k-means of 10 iterations, covariance, eigen-free 25x25 "
                        "solve (sweep inverse + repeated squaring), Mahalanobis max]); "
                        "latency- and residency-bound, not bandwidth-bound: the k-means parts "
                        "keep their points resident in LDS, so the live set is read ~3 times "
                        "per level, not 12"}
        if e2e is not None:
            line["config"]["end_to_end"] = e2e
        if cpu is not None:  # the CPU baseline is timed on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu
        os.write(record_fd, (json.dumps(line) + "\n").encode())
    dist.destroy_process_group()


def c3_leg(ctx, runs=16, queue=1024):
    from dynesty_amd import problems
    prob = problems.eggbox(2, name="C3")
    kw = dict(bound='multi', sample='rslice', slices=5, dlogz=0.01)
    ctx.ns_ensemble(prob, 2, 5000, queue, entropy=[3], max_fills=2, **kw)  # allocations, code objects
    t0 = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 5000, queue, entropy=[21], **kw)
    dt = time.perf_counter() - t0
    out = {"what": "eggbox 2-D, nlive 5000, MultiEllipsoid (13-15 ellipsoids), rslice x 5, device-resident loop",
           "runs": runs, "queue_size": queue, "seconds": dt, "seconds_per_run": dt / runs,
           "likelihood_calls_per_s": float(r["ncall"].sum() / dt), "status_ok": bool((r["status"] == 0).all()),
           "logz_mean": float(r["logz"].mean()), "logz_se": float(r["logz"].std(ddof=1) / math.sqrt(runs)),
           "logz_truth": 235.856, "mean_bound_updates": float(np.mean(r["nbound"]))}
    ref = os.path.join(ROOT, "tests", "golden", "c3_logz_ref.json")
    if os.path.exists(ref):
        ens = json.load(open(ref))["ensembles"]
        out["logz_reference"] = {k: {"mean": e["mean"], "se": e["se"], "n": e["n"],
                                     "mean_seconds_1core": e["mean_seconds_1core"]} for k, e in ens.items()}
    return out


def c1_leg(ctx, runs=64, queue=64):
    """BASELINE C1 (the reference's own CPU-runnable case) with the reference's defaults for sample='unif':
    bootstrap 5, enlarge 1 -- every rebuild of the loop runs 5 resampled replicas per run (boot.hip)."""
    from dynesty_amd import problems
    prob = problems.gauss_iid(3, 10.0, "C1")
    kw = dict(bound='single', sample='unif', dlogz=0.01)
    ctx.ns_ensemble(prob, 2, 500, queue, entropy=[3], max_fills=2, **kw)  # allocations, code objects
    t0 = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 500, queue, entropy=[21], **kw)
    dt = time.perf_counter() - t0
    out = {"what": "3-D unit Normal, prior +-10, nlive 500, single ellipsoid + bootstrap 5, uniform sampler, "
                   "device-resident loop",
           "runs": runs, "queue_size": queue, "seconds": dt, "seconds_per_run": dt / runs,
           "likelihood_calls_per_s": float(r["ncall"].sum() / dt), "status_ok": bool((r["status"] == 0).all()),
           "logz_mean": float(r["logz"].mean()), "logz_se": float(r["logz"].std(ddof=1) / math.sqrt(runs)),
           "logz_truth": float(prob.logz_truth), "mean_bound_updates": float(np.mean(r["nbound"]))}
    ref = os.path.join(ROOT, "tests", "golden", "c1_logz_ref.json")
    if os.path.exists(ref):
        ens = json.load(open(ref))["ensembles"]
        out["logz_reference"] = {k: {"mean": e["mean"], "se": e["se"], "n": e["n"],
                                     "mean_seconds_1core": e["mean_seconds_1core"]} for k, e in ens.items()
                                 if k.startswith("single")}
    return out


def c4_leg(ctx, runs=16, queue=128):
    """BASELINE C4 at the queue size whose ln Z stays within the gate of the SERIAL reference ensemble (ln Z drifts
    down with the queue size, in the reference as on the device: tests/test_gpu_logz_gate.py)."""
    from dynesty_amd import problems
    prob = problems.gauss_normal_prior(200, "C4")
    kw = dict(bound='single', sample='rslice', slices=203, dlogz=0.01, max_iter=250000)
    ctx.ns_ensemble(prob, 1, 4000, queue, entropy=[3], max_fills=2, **kw)  # allocations, code objects
    t0 = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 4000, queue, entropy=[21, queue], **kw)
    dt = time.perf_counter() - t0
    # the late form of the forced update (round 4's default; fewer bound updates than the reference) beside it
    t1 = time.perf_counter()
    rl = ctx.ns_ensemble(prob, runs, 4000, queue, entropy=[21, queue], forced_exact=False, **kw)
    dtl = time.perf_counter() - t1
    late = {"seconds": dtl, "logz_mean": float(rl["logz"].mean()), "logz_se": float(rl["logz"].std(ddof=1) / math.sqrt(runs)),
            "bound_updates_per_run": float(rl["nbound"].mean()), "fills": int(rl["nfills"])}
    out = {"what": "200-D iid Normal / Normal prior, nlive 4000, single ellipsoid, rslice x 203, device-resident loop",
           "protocol": "the reference's (forced_exact, default); forced_late = the late form",
           "bound_updates_per_run": float(r["nbound"].mean()), "fills": int(r["nfills"]), "forced_late": late,
           "runs": runs, "queue_size": queue, "seconds": dt, "seconds_per_run": dt / runs,
           "likelihood_calls_per_s": float(r["ncall"].sum() / dt), "status_ok": bool((r["status"] == 0).all()),
           "logz_mean": float(r["logz"].mean()), "logz_se": float(r["logz"].std(ddof=1) / math.sqrt(runs)),
           "logz_truth": float(prob.logz_truth)}
    ref = os.path.join(ROOT, "tests", "golden", "c4_logz_ref.json")
    if os.path.exists(ref):
        ens = json.load(open(ref)).get("ensembles", {})
        out["logz_reference"] = {k: {"mean": e["mean"], "se": e["se"], "n": e["n"],
                                     "mean_seconds_1core": e["mean_seconds_1core"]} for k, e in ens.items()}
    # roofline of the C4 loop as a whole (BASELINE.md section 3 asks for fp64 GFLOP/s here).
    # Algorithmic bytes: one F evaluation = read u, write u', write ln L = 8 (2 D + 1) = 3 208 B (slice
    # evaluation); flops: per slice one frame product of 2 D^2, per evaluation the prior transform + iid-Normal ln L
    # ~ 2 D + 3 D (the ndtri of the Normal prior is ~60 flop-equivalents per coordinate: counted apart, not as flops).
    # Both over the WHOLE loop's seconds (bound updates and queue consumption included: wide_walk_kernel is ~83 % of it,
    # profiles/r05); slices = queue entries walked (fills x K x runs) x slices per proposal, wasted entries included.
    d, slices = 200, 203
    ncall = float(r["ncall"].sum())
    nslices = float(r["nfills"]) * queue * runs * slices
    bytes_alg = ncall * 8 * (2 * d + 1)
    flops = nslices * 2.0 * d * d + ncall * 5.0 * d
    out["roofline"] = {"bound": "hbm", "achieved": bytes_alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": bytes_alg / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_evaluation": 8 * (2 * d + 1), "evaluations": ncall,
                       "evaluations_per_slice": ncall / nslices,
                       "fp64_gflops": flops / dt / 1e9, "fp64_frac_of_vector_peak": flops / dt / 1e12 / FP64_PEAK_TFLOPS,
                       "seconds": dt,
                       "note": "whole-loop figures (not one kernel): the walkers stay in LDS / registers for all 203 slices, "
                               "so the 3 208 B per evaluation never travel; what travels is the run's 320 KB frame, "
                               "streamed from L2 once per slice and workgroup (DESIGN.md 3.5)"}
    return out


def reference_logz_gate():
    """The real reference's C2 ensembles (tests/golden/c2_logz_ref.json, generated in the build
    container by tools/ref_c2_runs.py) at the queue size of the end-to-end leg."""
    p = os.path.join(ROOT, "tests", "golden", "c2_logz_ref.json")
    if not os.path.exists(p):
        return {"logz_reference_seed21": -57.4541}
    with open(p) as f:
        g = json.load(f)
    out = {"logz_reference_seed21_K1": -57.4541}
    for key, e in g.get("ensembles", {}).items():
        out[f"logz_reference_{key}"] = {"mean": e["mean"], "se": e["se"], "n": e["n"]}
    return out


def _profile_path(name):
    """The newest committed record of that name (profiles/r06/final = the round's last code, else r06, r05, r04)."""
    for rnd in (os.path.join("r06", "final"), "r06", "r05", "r04"):
        p = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r06", name)


def _load_profile(name):
    p = _profile_path(name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def tap_b_record():
    """Tap B (the real dynesty.NestedSampler over the drop-in surface) needs the reference package, which is on neither
    the GPU box nor in this repository: it was measured once on an MI355X box through a scratch copy
    (tools/tapb_hw.py) and is quoted from the committed record."""
    t = _load_profile("tapb_c2.json")
    if not t:
        return None
    runs = list(t["runs"].values())
    return {"source": os.path.relpath(_profile_path("tapb_c2.json"), ROOT) + " (tools/tapb_hw.py on an MI355X box through tools/stage_reference.sh; not re-measured in this run)",
            "what": t["what"], "runs": len(runs),
            "seconds_per_run": float(np.mean([r["seconds"] for r in runs])),
            "proposals_per_s": float(np.mean([r["proposals_per_s"] for r in runs])),
            "iterations_per_s": float(np.mean([r["iterations_per_s"] for r in runs])),
            "proposals_per_s_device_side": float(np.mean([r["proposals_per_s_device_side"] for r in runs])),
            "host_python_fraction": float(np.mean([r["host_python_seconds"] / r["seconds"] for r in runs])),
            "logz": [r["logz"] for r in runs]}


def _cpu_walk_worker(job):
    """One host core: oracle rwalk walkers for `budget_s` seconds."""
    seed, u0, loglstar, axes, scale, walks, budget_s = job
    sys.path.insert(0, ROOT)
    from oracle import proposals_ref as P
    prob = c2_problem()
    kids = np.random.SeedSequence(seed).spawn(200000)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        rng = np.random.Generator(np.random.PCG64(kids[n]))
        P.rwalk(u0[n % len(u0)].copy(), loglstar, axes, scale, prob.prior_transform,
                prob.loglikelihood, rng, walks)
        n += 1
    return n, time.perf_counter() - t0


def _reference_leg(prob, u0, nlive, scale, loglstar, walks, budget_s):
    """The REAL reference on one host core, when a copy of it is importable: /root/reference (the build container)
    or a staged scratch copy named by DYNESTY_REF_PY (never committed; recipe: tools/stage_reference.sh).  The same
    bounded sample as the port's: five MultiEllipsoid.update + scale_to_logvol(ln 1.25) of the shard's live sets
    (bounding.py:632-724), then RWalkSampler.sample (internal_samplers.py:504-561) on the shard's start points until
    the budget is spent.  Returns None when the reference is not there."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import refshim
        if not refshim.have_reference():
            return None
        dynesty = refshim.import_reference()
        from dynesty import bounding as RB
        from dynesty.internal_samplers import RWalkSampler, SamplerArgument
    except Exception:
        return None
    d = prob.ndim
    t_rebuilds, bound = [], None
    for r in range(5):
        pts = u0[r * nlive:(r + 1) * nlive] if len(u0) >= (r + 1) * nlive else u0[:nlive]
        bound = RB.MultiEllipsoid(d)
        t0 = time.perf_counter()
        bound.update(pts, rstate=np.random.default_rng(r))
        bound.scale_to_logvol(bound.logvol + math.log(1.25))
        t_rebuilds.append(time.perf_counter() - t0)
    t_rebuild = float(np.median(t_rebuilds))
    axes = bound.ells[0].axes
    kids = np.random.SeedSequence(99).spawn(200000)
    kw = dict(walks=walks)
    sample = u0[:4096]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        RWalkSampler.sample(SamplerArgument(u=sample[n % len(sample)].copy(), loglstar=loglstar, axes=axes, scale=scale,
                                            prior_transform=prob.prior_transform, loglikelihood=prob.loglikelihood,
                                            rseed=kids[n], kwargs=kw))
        n += 1
    dt = time.perf_counter() - t0
    per_prop = dt / (n * walks)
    return {"value": 1.0 / (per_prop + t_rebuild / (nlive * walks)), "unit": "proposals/s", "cores": 1,
            "kind": "reference", "rebuilds_per_s": 1.0 / t_rebuild, "proposals_per_s_walk_only": 1.0 / per_prop,
            "sample": f"5 x dynesty.bounding.MultiEllipsoid.update + scale_to_logvol of {nlive}x{d} live sets (median "
                      f"{t_rebuild * 1e3:.0f} ms, range {min(t_rebuilds) * 1e3:.0f}-{max(t_rebuilds) * 1e3:.0f} ms) + {n} x "
                      f"RWalkSampler.sample of {walks} steps ({dt:.1f} s): dynesty {getattr(dynesty, '__version__', '?')} "
                      f"from {refshim.REF}, 1 thread"}


def cpu_baseline(prob, u0, nlive, scale, loglstar, walks, budget_s):
    """The oracle (NumPy restatement of the reference's MultiEllipsoid.update +
    RWalkSampler.sample) timed on the host cores on a bounded sample of the same
    workload: one rebuild of run 0's live set, then walkers until the budget is
    spent; proposals/s is scaled to the reference cadence of one rebuild per
    nlive*walks proposals.  First on one core (the reference's serial path, the
    primary figure), then on all host cores as independent single-core workers
    (the ensemble mode of SURVEY.md section 8d)."""
    import multiprocessing as mp
    from oracle import bounding_ref as B
    from oracle import proposals_ref as _P  # noqa: F401  (imported here so that forked workers inherit it)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    # rebuild: the median of five (one per live set of the shard's first five runs) -- a single timing
    # swung 28 -> 106 ms between rounds on the GPU box's host
    t_rebuilds = []
    for r in range(5):
        pts = u0[r * nlive:(r + 1) * nlive] if len(u0) >= (r + 1) * nlive else u0[:nlive]
        t0 = time.perf_counter()
        mell = B.multi_update(pts)
        mell = B.scale_multi_to_logvol(mell, mell.logvol + math.log(1.25))
        t_rebuilds.append(time.perf_counter() - t0)
    t_rebuild = float(np.median(t_rebuilds))
    axes = mell.ells[0].axes
    sample = u0[:4096]
    n, dt = _cpu_walk_worker((99, sample, loglstar, axes, scale, walks, budget_s * 0.6))
    per_prop = dt / (n * walks)
    combined = 1.0 / (per_prop + t_rebuild / (nlive * walks))
    out = {"value": combined, "unit": "proposals/s", "cores": 1,
           "kind": "port",
           "rebuilds_per_s": 1.0 / t_rebuild,
           "proposals_per_s_walk_only": 1.0 / per_prop,
           "sample": f"5 rebuilds of {nlive}x{prob.ndim} live sets "
                     f"(median {t_rebuild * 1e3:.0f} ms, range {min(t_rebuilds) * 1e3:.0f}-{max(t_rebuilds) * 1e3:.0f} ms) "
                     f"+ {n} walkers x {walks} steps "
                     f"({dt:.1f} s), oracle/ (NumPy/SciPy restatement), 1 thread",
           "reference_figure": {
               "proposals_per_s": REFERENCE_C2_PROPOSALS_PER_S_1CORE,
               "rebuilds_per_s": REFERENCE_C2_REBUILDS_PER_S_1CORE, "cores": 1,
               "note": "the real dynesty 3.0.0 on C2, measured in the build container (SURVEY.md "
                       "section 6); it cannot travel to the GPU box, so the port above is what "
                       "is timed here -- the port is the faster of the two"}}
    ref_box = _load_profile("reference_cpu_on_gpu_box.json")
    if ref_box and ref_box.get("bounded_phase"):
        # the reference ITSELF on one host core of an MI355X box (round 4, through a scratch copy that is not part of
        # the repository: tools/tapb_hw.py) -- a committed measurement, not re-timed by this run
        out["reference_on_gpu_box"] = {
            "proposals_per_s": ref_box["bounded_phase"]["proposals_per_s"], "cores": 1,
            "multiellipsoid_update_ms": ref_box["multiellipsoid_update_ms"]["median"],
            "sample": ref_box["sample"], "host_cpu_count": ref_box.get("cpu_count"),
            "source": os.path.relpath(_profile_path("reference_cpu_on_gpu_box.json"), ROOT)}
    pool_box = _load_profile("reference_pool_on_gpu_box.json")
    if pool_box and pool_box.get("legs"):
        # the reference's OWN parallel path (dynesty.pool.Pool(ncores), queue_size = ncores) on the
        # host cores of an MI355X box (round 6, tools/ref_pool_hw.py through a staged copy) -- a committed measurement:
        # bound updates stay serial in the master process and proposals go through pool.map with chunksize 1
        out["reference_pool"] = {
            "unit": "proposals/s", "what": pool_box["what"], "host_cpu_count": pool_box.get("cpu_count"),
            "legs": [{"cores": l["cores"], "proposals_per_s": l.get("proposals_per_s"), "iterations_per_s": l.get("iterations_per_s")}
                     for l in pool_box["legs"]],
            "best": pool_box.get("best"),
            "source": os.path.relpath(_profile_path("reference_pool_on_gpu_box.json"), ROOT)}
    # the real reference, when this machine has a copy: it becomes the baseline, the port stays beside it
    ref = _reference_leg(prob, u0, nlive, scale, loglstar, walks, budget_s * 0.6)
    if ref is not None:
        port = {k: out[k] for k in ("value", "unit", "cores", "kind", "rebuilds_per_s", "proposals_per_s_walk_only", "sample")}
        out.update(ref)
        out["port"] = port
        out.pop("reference_figure", None)
    ncores = os.cpu_count() or 1
    if ncores > 1:
        try:
            mpc = mp.get_context("fork")  # safe: called before any HIP / torch initialisation
            with mpc.Pool(ncores) as pool:
                t0 = time.perf_counter()
                res = pool.map(_cpu_walk_worker,
                               [(1000 + i, sample, loglstar, axes, scale, walks, budget_s * 0.4)
                                for i in range(ncores)])
                wall = time.perf_counter() - t0
            rate = sum(r[0] for r in res) * walks / max(r[1] for r in res)
            out["all_cores"] = {
                "value": 1.0 / (1.0 / rate + t_rebuild / (nlive * walks) / ncores),
                "unit": "proposals/s", "cores": ncores,
                "sample": f"{ncores} independent single-core workers x {budget_s * 0.4:.1f} s of "
                          f"oracle rwalk walkers ({sum(r[0] for r in res)} walkers, pool wall "
                          f"{wall:.1f} s incl. start-up); rebuild cost spread over the cores"}
        except Exception as exc:  # the all-cores leg is auxiliary: never lose the line
            out["all_cores"] = {"error": repr(exc)}
    return out


if __name__ == "__main__":
    main()
