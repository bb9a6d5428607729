#!/usr/bin/env python
"""bench.py -- headline benchmark of the dynesty bounding + proposal hot path
on MI355X (contract: see the round prompt; metric from BASELINE.json).

Workload (config.workload): BASELINE config C2 -- 25-D rho=0.4 correlated
Normal, nlive=2000, bound='multi', sample='rwalk' (walks = 45) -- as `runs`
independent runs per GPU (the per-GPU shard of the C5 ensemble; 64 by default).
One *step* = one pass of the hot path for every run of the shard with all
inputs resident in HBM:  K = nlive walkers x `walks` rwalk proposals each
(in-kernel PCG64/ziggurat draws, frame mat-vec, prior transform, Gaussian
log-likelihood, accept test) = one bound-update interval of the reference
(update_interval = walks * nlive calls, dynesty.py:213-232).

value = proposals/s over all GPUs (weak scaling: per-GPU work fixed).
Tap point: (A) kernel boundary.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6  # vector fp64 (datasheet)


def make_shard(prob, runs, nlive, seed):
    """Synthetic live sets: points of a C2 posterior shell, their frame, and a
    likelihood threshold at the 10% quantile (so proposals are genuinely
    accepted/rejected)."""
    d = prob.ndim
    rng = np.random.default_rng(seed)
    cov = np.full((d, d), 0.4)
    np.fill_diagonal(cov, 1.0)
    lam, vec = np.linalg.eigh(cov)
    hw = prob.prior_par[0]
    # live points ~ N(0, s^2 cov) in v, mapped to the unit cube
    s = 0.6
    z = rng.standard_normal((runs * nlive, d))
    v = s * (z * np.sqrt(lam)) @ vec.T
    u0 = 0.5 + v / (2 * hw)
    logl = prob.loglikelihood_many(prob.prior_transform_many(u0))
    logl = logl.reshape(runs, nlive)
    loglstar = float(np.quantile(logl, 0.10))
    # bounding-ellipsoid-like frame in cube units: axes = V sqrt(lam) * r
    radius = s * math.sqrt(d + 2 * math.sqrt(2 * d)) / (2 * hw) * 1.08
    axes = (vec * np.sqrt(lam)) * radius
    return u0, axes, loglstar


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--runs", type=int, default=64,
                    help="independent C2 runs per GPU (C5 shard = 64)")
    ap.add_argument("--nlive", type=int, default=2000)
    ap.add_argument("--walks", type=int, default=45)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from dynesty_amd import _lib, problems
    ctx = _lib.Context(local_rank)
    lib = ctx.lib
    prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
    d = prob.ndim
    k = args.runs * args.nlive
    u0, axes, loglstar = make_shard(prob, args.runs, args.nlive, 1000 + rank)
    scale = 0.35

    # resident device buffers
    h = ctx.handle
    d_u0 = ctx.to_device(u0)
    d_axes = ctx.to_device(axes)
    ent = [21, rank, 0, 0]
    states = ctx.seed_children(ent, 0, k)
    d_rng = ctx.to_device(states)
    d_u = ctx.malloc(k * d * 8)
    d_v = ctx.malloc(k * d * 8)
    d_logl = ctx.malloc(k * 8)
    d_na = ctx.malloc(k * 4)
    d_nr = ctx.malloc(k * 4)
    d_rng_out = ctx.malloc(k * 32)
    ph = ctx.problem(prob)

    def step(rng_in, rng_out):
        rc = lib.dh_rwalk_batch_dev(h, ph, k, d, d, d_u0, d_axes, 1, None,
                                    scale, loglstar, args.walks, None, rng_in,
                                    d_u, d_v, d_logl, d_na, d_nr, rng_out)
        ctx._check(rc)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for i in range(args.warmup):
        step(d_rng if i % 2 == 0 else d_rng_out, d_rng_out if i % 2 == 0 else d_rng)
    barrier()
    ev0, ev1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(ev0)
    for i in range(args.steps):
        # streams continue from step to step (ping-pong the state buffers)
        step(d_rng if i % 2 == 0 else d_rng_out, d_rng_out if i % 2 == 0 else d_rng)
    ctx.record(ev1)
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    kern_ms = ctx.elapsed_ms(ev0, ev1) / args.steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    nacc = ctx.from_device(d_na, (k,), np.int32)
    nrej = ctx.from_device(d_nr, (k,), np.int32)
    assert np.all(nacc + nrej == args.walks)
    props_per_step_rank = k * args.walks
    value = world * props_per_step_rank * args.steps / wall

    # ensemble exchange step (C5): gather one record per run over RCCL
    if dist is not None:
        rec = torch.tensor(nacc.reshape(args.runs, -1).mean(1), device="cuda")
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)

    if rank == 0:
        alg_bytes = 8 * (2 * d + 1)  # read u, write u', write logl
        flops = 2 * d * d + 8 * d + (d * d + 3 * d)  # frame mat-vec + sym. quad form
        achieved = props_per_step_rank * alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "proposals/sec (25-D corr-Normal nlive=2000 multi/rwalk)",
            "value": value,
            "unit": "proposals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2 x {args.runs} independent runs per GPU "
                            f"(C5 shard): K={args.nlive} walkers x "
                            f"{args.walks} rwalk steps per run per step",
                "ndim": d, "nlive": args.nlive, "walks": args.walks,
                "runs_per_gpu": args.runs, "tap_point": "A (kernel boundary)",
                "accept_frac": float(nacc.sum() / (k * args.walks)),
            },
            "roofline": {
                "bound": "hbm", "kernel": "rwalk_kernel<25,true>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel_ms": kern_ms,
                "fp64_valu": {
                    "achieved": props_per_step_rank * flops / (kern_ms * 1e-3) / 1e12,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s"},
            },
        }
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(prob, u0, axes, scale, loglstar,
                                                args.walks, args.cpu_seconds)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(prob, u0, axes, scale, loglstar, walks, budget_s):
    """The oracle (NumPy restatement of the reference's RWalkSampler.sample)
    timed on one host core on a bounded sample of the same workload."""
    from oracle import proposals_ref as P
    kids = np.random.SeedSequence(99).spawn(100000)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        rng = np.random.Generator(np.random.PCG64(kids[n]))
        P.rwalk(u0[n].copy(), loglstar, axes, scale, prob.prior_transform,
                prob.loglikelihood, rng, walks)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * walks / dt, "unit": "proposals/s", "cores": 1,
            "kind": "port",
            "sample": f"{n} walkers x {walks} steps of the same C2 shard "
                      f"({dt:.1f} s, oracle/proposals_ref.rwalk, 1 thread)"}


if __name__ == "__main__":
    main()
