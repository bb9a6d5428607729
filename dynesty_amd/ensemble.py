"""Many independent runs sharded over the GPUs of one node (BASELINE config C5).

Runs share nothing: rank r -- one process per GPU,
launched by torchrun / torch.distributed.run -- owns a contiguous block of run
ids and executes them on its own device with no communication.  Seeds are
derived from the *global* run id (``SeedSequence(base).spawn(total)[run]``), so
results do not depend on the number of GPUs.  The only exchange step is at the
end: one fixed-size record per run, all-gathered with RCCL over xGMI (backend
"nccl" on ROCm; "gloo" in the CPU tests).  At 512 x 5 doubles = 20 KB the
collective is latency bound; posterior samples stay sharded (they can be merged
on the host with the reference's ``utils.merge_runs``) unless ``gather_ragged``
is asked for them.
"""
import numpy as np

RECORD_FIELDS = ("run", "logz", "logzerr", "niter", "ncall", "h")


def shard_runs(total, world, rank):
    """Contiguous block of run ids owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def run_seeds(base_seed, total):
    """One SeedSequence child per global run id."""
    return np.random.SeedSequence(base_seed).spawn(int(total))


def gather_records(local, total, world, rank, dist=None, device=None, nfield=None):
    """All-gather the per-run records (n_local x len(RECORD_FIELDS), float64;
    column 0 is the global run id).  Returns the (total x nfield) table ordered
    by run id on every rank.  With dist=None (no process group) it is a sort;
    with a process group the collective runs for every world size, 1 included
    (a one-rank RCCL communicator is still RCCL: `bench.py` and the `-m gpu`
    tests go through it on a single MI355X)."""
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(
        -1, nfield or len(RECORD_FIELDS))
    if dist is None:
        out = local
    else:
        import torch
        nmax = -(-int(total) // int(world))  # ceil: equal-sized buffers
        buf = np.full((nmax, local.shape[1]), np.nan)
        buf[:len(local)] = local
        t = torch.from_numpy(buf)
        if device is not None:
            t = t.to(device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        out = np.concatenate([p.cpu().numpy() for p in parts])
        out = out[~np.isnan(out[:, 0])]
    out = out[np.argsort(out[:, 0], kind="stable")]
    assert len(out) == total, (len(out), total)
    return out


def gather_ragged(arrays, world, rank, dist=None, device=None):
    """Optional second exchange: gather variable-length (n_i, d) float64 arrays
    (e.g. weighted posterior samples) as [counts all-gather, padded
    all-gather].  Returns the list of arrays of every rank, in rank order."""
    arrays = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    if dist is None:
        return arrays
    import torch
    d = arrays[0].shape[1] if arrays else 0
    flat = np.concatenate(arrays) if arrays else np.zeros((0, d))
    counts = torch.tensor([len(a) for a in arrays] or [0], dtype=torch.int64)
    meta = torch.tensor([len(arrays), len(flat), d], dtype=torch.int64)
    if device is not None:
        meta = meta.to(device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().numpy() for m in metas]
    max_runs = max(int(m[0]) for m in metas) or 1
    max_rows = max(int(m[1]) for m in metas) or 1
    d = max(int(m[2]) for m in metas)
    cbuf = torch.zeros(max_runs, dtype=torch.int64)
    cbuf[:len(arrays)] = counts[:len(arrays)]
    fbuf = torch.zeros((max_rows, d), dtype=torch.float64)
    if len(flat):
        fbuf[:len(flat)] = torch.from_numpy(flat)
    if device is not None:
        cbuf, fbuf = cbuf.to(device), fbuf.to(device)
    call = [torch.empty_like(cbuf) for _ in range(world)]
    fall = [torch.empty_like(fbuf) for _ in range(world)]
    dist.all_gather(call, cbuf)
    dist.all_gather(fall, fbuf)
    out = []
    for r in range(world):
        c = call[r].cpu().numpy()[:int(metas[r][0])]
        f = fall[r].cpu().numpy()
        off = 0
        for n in c:
            out.append(f[off:off + int(n)].copy())
            off += int(n)
    return out


def run_ensemble(prob, total_runs, base_seed=21, world=1, rank=0, dist=None,
                 device=None, **run_kw):
    """Run this rank's shard with ``nested.run_static`` and gather the records.
    Returns (table, local_results)."""
    from . import nested
    seeds = run_seeds(base_seed, total_runs)
    mine = shard_runs(total_runs, world, rank)
    local, results = [], []
    for rid in mine:
        rng = np.random.Generator(np.random.PCG64(seeds[rid]))
        r = nested.run_static(prob, rstate=rng, **run_kw)
        results.append(r)
        local.append([rid, r.logz, r.logzerr, r.niter, r.ncall, r.h])
    table = gather_records(np.array(local).reshape(-1, len(RECORD_FIELDS)),
                           total_runs, world, rank, dist=dist, device=device)
    return table, results


STATUS_RAISED = -99.0  # status of the runs of a rank whose ns_ensemble call raised (never a device status)


def run_ensemble_device(prob, total_runs, base_seed=21, world=1, rank=0,
                        dist=None, device=None, nlive=2000, queue_size=512,
                        walks=None, bound='multi', dlogz=0.01, on_failure='raise', **kw):
    """BASELINE config C5: this rank's shard of `total_runs` static runs executed
    by the device-resident loop (`dh_ns_ensemble`, one launch sequence for the
    whole shard), then the RCCL all-gather of the per-run records.  Seeds are
    keyed on the global run id, so the table does not depend on `world`.
    on_failure: 'raise' (default) or 'nan' -- what to do with a run whose status is
    not 0; with 'nan' its ln Z / error / information are NaN in the table and
    `combine_logz` leaves it out."""
    if on_failure not in ('raise', 'nan'):
        raise ValueError("on_failure must be 'raise' or 'nan'")
    from .backend import get_backend
    mine = shard_runs(total_runs, world, rank)
    nf = len(RECORD_FIELDS)
    local = np.zeros((0, nf + 1))
    raised = None
    if len(mine):
        try:
            r = get_backend().ns_ensemble(prob, len(mine), nlive, queue_size,
                                          walks=walks, bound=bound, dlogz=dlogz,
                                          entropy=np.atleast_1d(base_seed),
                                          first_run=mine.start, **kw)
        except Exception as e:  # argument / memory / HIP errors raise before any status exists
            #  this rank must still enter the collective, else the others block in the all-gather
            # until the communicator times out: its runs travel as NaN rows with the sentinel status, and every
            # rank raises after the gather; this rank re-raises its own exception
            raised = e
            r = None
        if r is None:
            local = np.full((len(mine), nf + 1), np.nan)
            local[:, 0] = np.arange(mine.start, mine.stop)
            local[:, nf] = STATUS_RAISED
        else:
            status = np.asarray(r["status"]).astype(np.float64)
            bad = np.flatnonzero(status != 0)
            # a failed run (bound rebuild error, dead-point capacity) or one that hit max_fills
            # before dlogz must never enter the table as a valid ln Z estimate
            for key in ("logz", "logzerr", "h"):
                r[key] = np.array(r[key], dtype=np.float64)
                r[key][bad] = np.nan
            local = np.stack([np.arange(mine.start, mine.stop, dtype=np.float64),
                              r["logz"], r["logzerr"], r["niter"].astype(float),
                              r["ncall"].astype(float), r["h"], status], axis=1)
    # The collective comes FIRST: the status column travels with the records, so that every rank
    # learns of a failed run and all of them raise together -- raising on the owning rank alone
    # would leave the others blocked in the all-gather until the communicator times out.
    table = gather_records(local, total_runs, world, rank, dist=dist,
                           device=device, nfield=nf + 1)
    status = table[:, nf]
    table = np.ascontiguousarray(table[:, :nf])
    if raised is not None:
        raise raised
    if np.any(status == STATUS_RAISED):
        raise RuntimeError(
            f"ns_ensemble raised on the rank(s) owning runs "
            f"{[int(table[b, 0]) for b in np.flatnonzero(status == STATUS_RAISED)]}")
    bad = np.flatnonzero(status != 0)
    if len(bad) and on_failure == 'raise':
        raise RuntimeError(
            f"ns_ensemble: runs {[int(table[b, 0]) for b in bad]} ended with status "
            f"{[int(status[b]) for b in bad]} (1 = not converged within max_fills, "
            f"< 0 = failed)")
    return table


def combine_logz(table):
    """Ensemble estimate from the gathered record table: mean and standard error of the
    per-run ln Z over the runs that completed (failed runs carry NaN, see
    run_ensemble_device(on_failure='nan')).  Returns (mean, se, n_used)."""
    lz = np.asarray(table)[:, RECORD_FIELDS.index("logz")]
    ok = np.isfinite(lz)
    n = int(ok.sum())
    if n == 0:
        raise RuntimeError("combine_logz: no completed run in the table")
    se = float(lz[ok].std(ddof=1) / np.sqrt(n)) if n > 1 else float('nan')
    return float(lz[ok].mean()), se, n


class MergedRun(dict):
    """Result of `merge_static_runs`: the fields of the reference's merged
    `Results` (utils.merge_runs) that are defined for an ensemble of static
    runs, with attribute access."""
    __getattr__ = dict.get

    def importance_weights(self):
        """Normalised posterior weights exp(logwt - logz[-1])."""
        w = np.exp(self["logwt"] - self["logz"][-1])
        return w / w.sum()


def merge_static_runs(dead_logl, niter, live_logl, dead_u=None, live_u=None,
                      prior_transform=None, ncall=None, dead_id=None, dead_it=None,
                      dead_nc=None, live_it=None, live_id=None):
    """Combine R static runs of equal, constant nlive into ONE run with R*nlive
    live points -- what the reference's utils.merge_runs / _merge_two
    (utils.py:1817-1900, 2000-2226) produce for such runs.

    Every point (a run's dead points in death order, then its final live points
    in ascending log-likelihood) carries the number of live points its own run
    had when it died (N for dead points; N, N-1, ..., 1 for the final ones);
    the merged sequence is ordered by log-likelihood (stable) and each point
    shrinks the prior volume by n/(n+1) with n the summed live count; weights,
    ln Z, information and var[ln Z] follow utils.compute_integrals
    (utils.py:1411-1467).

    dead_logl: (R, >=max niter), niter: (R,), live_logl: (R, N) in slot order;
    dead_u: (R, >=max niter, D) and live_u: (R, N, D) optional (posterior
    samples); prior_transform: optional callable mapping an (n, D) array of
    unit-cube points to parameters (gives `samples`).

    dead_id / dead_it / dead_nc: (R, >=max niter) and live_it: (R, N) optional
    (together): the per-point bookkeeping the reference's runs carry -- live
    slot, iteration at which the point was proposed, likelihood calls spent on
    its replacement (sampler.py:1165-1182; the final live points have id = their
    slot and nc = 1, sampler.py:870-890) -- which _merge_two copies point by
    point into samples_id / samples_it / ncall (utils.py:2154-2156, 2196-2207).
    live_id: (R, N) slots of the rows of live_logl when they are not in slot order.

    Returns a MergedRun with niter, logl, logvol, logwt, logz, logzerr,
    information (all per point, cumulative where the reference's are),
    samples_n, samples_run (run index of each point), samples_seq (index within
    its run's own sequence), with coordinates given samples_u / samples, and with
    the per-point bookkeeping given samples_id, samples_it and ncall (array; its
    sum replaces the `ncall` argument in eff = 100 niter / sum(ncall))."""
    from .nested import _integrate_full
    dead_logl = np.asarray(dead_logl)
    live_logl = np.asarray(live_logl)
    R, N = live_logl.shape
    ls, ns, rs, its, us = [], [], [], [], []
    point_info = dead_id is not None
    ids, bits, ncs = [], [], []
    for r in range(R):
        k = int(niter[r])
        d = dead_logl[r, :k]
        lo = np.argsort(live_logl[r], kind="stable")
        ls.append(np.concatenate([d, live_logl[r][lo]]))
        # change of this run's live count AFTER each of its points dies
        ns.append(np.concatenate([np.zeros(k), -np.ones(N)]))
        rs.append(np.full(k + N, r, dtype=np.int64))
        its.append(np.arange(k + N, dtype=np.int64))
        if dead_u is not None:
            us.append(np.concatenate([np.asarray(dead_u[r][:k]),
                                      np.asarray(live_u[r])[lo]]))
        if point_info:
            ids.append(np.concatenate([np.asarray(dead_id[r][:k], dtype=np.int64),
                                       lo if live_id is None else np.asarray(live_id[r], dtype=np.int64)[lo]]))
            bits.append(np.concatenate([np.asarray(dead_it[r][:k], dtype=np.int64),
                                        np.asarray(live_it[r], dtype=np.int64)[lo]]))
            ncs.append(np.concatenate([np.asarray(dead_nc[r][:k], dtype=np.int64),
                                       np.ones(N, dtype=np.int64)]))
    logl = np.concatenate(ls)
    dn = np.concatenate(ns)
    order = np.argsort(logl, kind="stable")
    logl, dn = logl[order], dn[order]
    nlive_at = R * N + np.concatenate([[0.], np.cumsum(dn)[:-1]])
    logvol = -np.cumsum(np.log((nlive_at + 1.) / nlive_at))
    logwt, logz, h, logzvar = _integrate_full(logl, logvol)
    out = MergedRun(niter=len(logl), logl=logl, logvol=logvol, logwt=logwt,
                    logz=logz, logzerr=np.sqrt(logzvar), information=h,
                    samples_n=nlive_at.astype(np.int64),
                    samples_run=np.concatenate(rs)[order],
                    samples_seq=np.concatenate(its)[order])
    if point_info:
        out["samples_id"] = np.concatenate(ids)[order]
        out["samples_it"] = np.concatenate(bits)[order]
        out["ncall"] = np.concatenate(ncs)[order]
        out["eff"] = 100. * len(logl) / float(out["ncall"].sum())
    elif ncall is not None:
        out["ncall"] = int(np.sum(ncall))
        out["eff"] = 100. * len(logl) / out["ncall"]
    if us:
        out["samples_u"] = np.concatenate(us)[order]
        if prior_transform is not None:
            out["samples"] = np.asarray(prior_transform(out["samples_u"]))
    return out


def merge_logz(dead_logl, niter, live_logl):
    """(logz, logzerr) of the merged run (see merge_static_runs)."""
    m = merge_static_runs(dead_logl, niter, live_logl)
    return float(m["logz"][-1]), float(m["logzerr"][-1])


def run_ensemble_merged(prob, runs, nlive=2000, queue_size=512, entropy=(21,),
                        max_iter=None, **kw):
    """`runs` static runs on the device (dh_ns_ensemble) merged into one
    MergedRun with posterior samples: the single-process form of BASELINE C5's
    "gather of logZ / posterior samples"."""
    from .backend import get_backend
    be = get_backend()
    if max_iter is None:
        max_iter = 80 * nlive  # > nlive * (H + ln(1/dlogz)) for the benchmark problems
    r = be.ns_ensemble(prob, runs, nlive, queue_size, entropy=entropy,
                       max_iter=max_iter, want_samples=True, **kw)
    if (r["status"] != 0).any():
        raise RuntimeError(f"ns_ensemble: runs failed, status {r['status']}")

    def ptform(u):
        return be.problem_eval(prob, u)[0]
    m = merge_static_runs(r["dead_logl"], r["niter"], r["live_logl"],
                          r["dead_u"], r["live_u"], prior_transform=ptform,
                          dead_id=r["dead_id"], dead_it=r["dead_it"], dead_nc=r["dead_nc"],
                          live_it=r["live_it"])
    m["runs"] = r
    return m


def gather_and_merge(points_per_run, nlive, world=1, rank=0, dist=None,
                     device=None, prior_transform=None, ncall=None):
    """The sharded form of the combiner (north star: "gather of logZ / posterior
    samples"): every rank contributes, per local run, ONE array of rows
    [logl, is_final_live, id, it, nc, u_0 .. u_{D-1}] (`run_rows`: dead points in
    death order, then the final live points in any order; id = -1: no per-point
    bookkeeping); the rows travel in the ragged all-gather
    (`gather_ragged`: counts + padded rows, RCCL on GPUs, gloo on CPU) and every
    rank merges all runs with `merge_static_runs`.  Returns the MergedRun."""
    allruns = gather_ragged(points_per_run, world, rank, dist=dist, device=device)
    R = len(allruns)
    d = allruns[0].shape[1] - 5
    nit = np.array([int((a[:, 1] == 0).sum()) for a in allruns])
    dead_l = np.zeros((R, max(1, nit.max())))
    dead_u = np.zeros((R, max(1, nit.max()), d))
    dead_i = np.zeros((3, R, max(1, nit.max())), dtype=np.int64)
    live_l = np.zeros((R, nlive))
    live_u = np.zeros((R, nlive, d))
    live_i = np.zeros((2, R, nlive), dtype=np.int64)
    for i, a in enumerate(allruns):
        dmask = a[:, 1] == 0
        dead_l[i, :nit[i]] = a[dmask, 0]
        dead_u[i, :nit[i]] = a[dmask, 5:]
        dead_i[:, i, :nit[i]] = a[dmask, 2:5].T
        live_l[i] = a[~dmask, 0]
        live_u[i] = a[~dmask, 5:]
        live_i[:, i] = a[~dmask, 2:4].T
    info = {}
    if all((a[:, 2] >= 0).all() for a in allruns):
        info = dict(dead_id=dead_i[0], dead_it=dead_i[1], dead_nc=dead_i[2], live_id=live_i[0], live_it=live_i[1])
    if ncall is not None and dist is not None:
        import torch
        t = torch.tensor([float(np.sum(ncall))], dtype=torch.float64)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        ncall = [float(t.item())]
    return merge_static_runs(dead_l, nit, live_l, dead_u, live_u,
                             prior_transform=prior_transform, ncall=ncall, **info)


def run_rows(dead_logl, dead_u, live_logl, live_u, dead_id=None, dead_it=None, dead_nc=None, live_it=None):
    """One run as the row block `gather_and_merge` expects (live points in slot order)."""
    k, n = len(dead_logl), len(live_logl)
    out = np.empty((k + n, 5 + np.shape(live_u)[1]))
    out[:k, 0], out[k:, 0] = dead_logl, live_logl
    out[:k, 1], out[k:, 1] = 0., 1.
    if dead_id is None:
        out[:, 2:5] = -1.
    else:
        out[:k, 2], out[k:, 2] = dead_id, np.arange(n)
        out[:k, 3], out[k:, 3] = dead_it, live_it
        out[:k, 4], out[k:, 4] = dead_nc, 1.
    out[:k, 5:], out[k:, 5:] = dead_u, live_u
    return out


def run_ensemble_merged_sharded(prob, total_runs, base_seed=21, world=1, rank=0,
                                dist=None, device=None, nlive=2000,
                                queue_size=512, max_iter=None, **kw):
    """BASELINE C5 end to end: this rank's shard of `total_runs` device-resident
    runs (dh_ns_ensemble with coordinates), the ragged gather of every run's
    points, and the merged posterior / evidence on every rank."""
    from .backend import get_backend
    be = get_backend()
    mine = shard_runs(total_runs, world, rank)
    if max_iter is None:
        max_iter = 80 * nlive
    rows, ncall, nbad = [], 0, 0
    r, raised = None, None
    if len(mine):
        try:
            r = be.ns_ensemble(prob, len(mine), nlive, queue_size,
                               entropy=np.atleast_1d(base_seed), first_run=mine.start,
                               max_iter=max_iter, want_samples=True, **kw)
            nbad = int((r["status"] != 0).sum())
        except Exception as e:  # this rank still joins the all-reduce below, then re-raises
            raised = e
            nbad = len(mine)
    # every rank learns of a failure before anyone leaves the collective sequence (see
    # run_ensemble_device): one all-reduce of the failure count, then raise everywhere
    if dist is not None:
        import torch
        t = torch.tensor([float(nbad)], dtype=torch.float64)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        nbad_all = int(t.item())
    else:
        nbad_all = nbad
    if raised is not None:
        raise raised
    if nbad_all:
        raise RuntimeError(f"ns_ensemble: {nbad_all} run(s) of the ensemble failed"
                           + (f", local status {r['status']}" if nbad else ""))
    if len(mine):
        ncall = int(r["ncall"].sum())
        for i in range(len(mine)):
            k = int(r["niter"][i])
            rows.append(run_rows(r["dead_logl"][i, :k], r["dead_u"][i, :k],
                                 r["live_logl"][i], r["live_u"][i], r["dead_id"][i, :k],
                                 r["dead_it"][i, :k], r["dead_nc"][i, :k], r["live_it"][i]))

    def ptform(u):
        return be.problem_eval(prob, u)[0]
    return gather_and_merge(rows, nlive, world, rank, dist=dist, device=device,
                            prior_transform=ptform, ncall=[ncall])
