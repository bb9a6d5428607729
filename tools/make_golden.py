#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (dynesty 3.0.0
from /root/reference/py, numpy/scipy of this image) on the seeded inputs of
tests/inputs.py.  Run in the build container only:

    python tools/make_golden.py

The reference cannot travel to the GPU box, the fixtures can.  Nothing here is
imported by the product.  Import shim per SURVEY.md appendix B.
"""
import os
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_reference():
    shim = tempfile.mkdtemp(prefix="dynesty_shim_")
    di = os.path.join(shim, "dynesty-3.0.0.dist-info")
    os.makedirs(di)
    with open(os.path.join(di, "METADATA"), "w") as f:
        f.write("Metadata-Version: 2.1\nName: dynesty\nVersion: 3.0.0\n")
    sys.path.insert(0, shim)
    sys.path.insert(0, "/root/reference/py")
    import dynesty  # noqa
    return dynesty


def ell_fields(e):
    return dict(ctr=e.ctr, cov=e.cov, am=e.am, axes=e.axes, axlens=e.axlens,
                logvol=np.float64(e.logvol))


def gen_bounding(out):
    from dynesty import bounding as db
    import inputs
    g = {}
    for name in inputs.CLOUDS_SMALL:
        pts = inputs.cloud(name)
        n, d = pts.shape
        e = db.bounding_ellipsoid(pts)
        for k, v in ell_fields(e).items():
            g[f"{name}/be/{k}"] = v
        m = db.MultiEllipsoid(d)
        m.update(pts, rstate=np.random.default_rng(5))
        g[f"{name}/mu/nells"] = np.int64(m.nells)
        g[f"{name}/mu/ctrs"] = m.ctrs
        g[f"{name}/mu/covs"] = m.covs
        g[f"{name}/mu/ams"] = m.ams
        g[f"{name}/mu/logvol_ells"] = m.logvol_ells
        g[f"{name}/mu/logvol"] = np.float64(m.logvol)
        g[f"{name}/mu/axes"] = np.array([el.axes for el in m.ells])
        g[f"{name}/mu/axlens"] = np.array([el.axlens for el in m.ells])
        # which ellipsoid(s) each input point falls in: exact integer KAT
        # (reference: tests/test_ellipsoid.py:106-133 test_overlap)
        probe_rng = np.random.default_rng(11)
        probes = pts[probe_rng.integers(n, size=64)] + \
            0.3 * pts.std(axis=0) * probe_rng.standard_normal((64, d))
        wl = [m.within(p) for p in probes]
        g[f"{name}/kat/probes"] = probes
        g[f"{name}/kat/within_flat"] = np.concatenate(wl).astype(np.int64) \
            if len(wl) else np.zeros(0, np.int64)
        g[f"{name}/kat/within_count"] = np.array([len(w) for w in wl],
                                                 dtype=np.int64)
        g[f"{name}/kat/contains"] = np.array([m.contains(p) for p in probes])
        g[f"{name}/kat/overlap_skip0"] = np.array(
            [m.overlap(p, j=0) for p in probes], dtype=np.int64)
        # union samples, same seed (reference: bounding.py:592-606)
        g[f"{name}/mu/samples"] = m.samples(40, rstate=np.random.default_rng(7))
        # enlargement as applied by Sampler.update_bound (sampler.py:506-508)
        m.scale_to_logvol(m.logvol + np.log(1.25))
        g[f"{name}/sc/ctrs"] = m.ctrs
        g[f"{name}/sc/covs"] = m.covs
        g[f"{name}/sc/ams"] = m.ams
        g[f"{name}/sc/logvol_ells"] = m.logvol_ells
        g[f"{name}/sc/logvol"] = np.float64(m.logvol)
        g[f"{name}/sc/axes"] = np.array([el.axes for el in m.ells])
        g[f"{name}/sc/axlens"] = np.array([el.axlens for el in m.ells])
        # single-ellipsoid bound
        s = db.Ellipsoid(d)
        s.update(pts, rstate=np.random.default_rng(5))
        g[f"{name}/single/samples"] = s.samples(
            40, rstate=np.random.default_rng(8))
        g[f"{name}/single/contains"] = np.array(
            [s.contains(p) for p in probes])
        # anisotropic branch of scale_to_logvol (bounding.py:257-275):
        # ask for a volume that forces axes against the sqrt(D)/2 cap
        big = db.bounding_ellipsoid(pts)
        target = big.logvol + d * np.log(
            (np.sqrt(d) / 2) / big.axlens.max()) + 0.3 * d
        maxlv = d * np.log(np.sqrt(d) / 2) + db.logvol_prefactor(d)
        target = min(target, maxlv - 1e-3)
        big.scale_to_logvol(target)
        for k, v in ell_fields(big).items():
            g[f"{name}/aniso/{k}"] = v
        g[f"{name}/aniso/target"] = np.float64(target)
        # bootstrap expansion factors (bounding.py:1619-1648)
        seeds = np.random.SeedSequence(77).spawn(3)
        g[f"{name}/boot/single"] = np.array([
            db._ellipsoid_bootstrap_expand((False, pts, sd)) for sd in seeds])
        seeds = np.random.SeedSequence(77).spawn(3)
        g[f"{name}/boot/multi"] = np.array([
            db._ellipsoid_bootstrap_expand((True, pts, sd)) for sd in seeds])
    # improve_covar_mat on the reference's own edge cases
    # (tests/test_ellipsoid.py:242-255 test_bounds)
    for tag, mat in [("zero", np.zeros((4, 4))),
                     ("rank1", np.outer(np.arange(1., 5.), np.arange(1., 5.))),
                     ("neg", -np.eye(3))]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            good, cov, am, axes = db.improve_covar_mat(mat)
        g[f"icm/{tag}/in"] = mat
        g[f"icm/{tag}/good"] = np.bool_(good)
        g[f"icm/{tag}/cov"] = cov
        g[f"icm/{tag}/am"] = am
        g[f"icm/{tag}/axes"] = axes
    for nd in (1, 2, 3, 25, 200):
        g[f"prefactor/{nd}"] = np.float64(db.logvol_prefactor(nd))
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


def gen_proposals(out):
    from dynesty import internal_samplers as dis
    from dynesty import bounding as db
    from dynesty.utils import get_nonbounded
    import inputs
    g = {}

    def run(tag, cls, kwargs, case, nwalk, seedbase, scale=None, extra=None):
        prob = case["problem"]
        seeds = np.random.SeedSequence(seedbase).spawn(nwalk)
        smp = cls(**kwargs)
        kw = dict(smp.sampler_kwargs)
        if extra:
            kw.update(extra)
        res = []
        for i in range(nwalk):
            arg = dis.SamplerArgument(
                u=case["u0"][i].copy(), loglstar=case["loglstar"],
                axes=case["axes"], scale=scale or case["scale"],
                prior_transform=prob.prior_transform,
                loglikelihood=prob.loglikelihood, rseed=seeds[i], kwargs=kw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res.append(cls.sample(arg))
        g[f"{tag}/u"] = np.array([r.u for r in res])
        g[f"{tag}/v"] = np.array([r.v for r in res])
        g[f"{tag}/logl"] = np.array([float(r.logl) for r in res])
        g[f"{tag}/ncalls"] = np.array([r.ncalls for r in res], dtype=np.int64)
        ti = [r.tuning_info for r in res]
        if ti[0] is not None:
            for key in ti[0]:
                g[f"{tag}/ti_{key}"] = np.array([t[key] for t in ti])
        g[f"{tag}/seedbase"] = np.int64(seedbase)
        g[f"{tag}/nwalk"] = np.int64(nwalk)

    # ---- rwalk (internal_samplers.py:866-986) ----
    for pname, nw, walks in (("C2", 8, 45), ("G5", 16, 25), ("C1", 8, 23),
                             ("C3", 8, 22), ("N6", 8, 26)):
        case = inputs.walker_case(pname, 64, 900 + walks)
        d = case["problem"].ndim
        run(f"rwalk/{pname}", dis.RWalkSampler,
            dict(ndim=d, ncdim=d, walks=walks, nonbounded=None, periodic=None,
                 reflective=None), case, nw, 4000 + walks)
    # periodic + reflective dims
    case = inputs.walker_case("G5", 64, 931, shrink=3.0)
    per, ref = np.array([0, 3]), np.array([1])
    run("rwalk/G5_pr", dis.RWalkSampler,
        dict(ndim=5, ncdim=5, walks=30,
             nonbounded=get_nonbounded(5, per, ref), periodic=per,
             reflective=ref), case, 16, 4100, scale=2.5)
    # ncdim < ndim: last two dims are redrawn U(0,1) at every step
    case = inputs.walker_case("G5", 64, 932, shrink=2.0)
    case3 = dict(case)
    case3["axes"] = case["axes"][:3, :3].copy()
    run("rwalk/G5_nc3", dis.RWalkSampler,
        dict(ndim=5, ncdim=3, walks=20, nonbounded=None, periodic=None,
             reflective=None), case3, 16, 4200)

    # ---- rslice / slice (internal_samplers.py:745-855, 593-709, 1075-1206) --
    for pname, nw, slices in (("C3", 12, 5), ("G5", 12, 4), ("N6", 8, 3),
                              ("C2", 4, 3)):
        case = inputs.walker_case(pname, 64, 950 + slices)
        d = case["problem"].ndim
        run(f"rslice/{pname}", dis.RSliceSampler,
            dict(ndim=d, ncdim=d, slices=slices, nonbounded=None,
                 periodic=None, reflective=None), case, nw, 5000 + slices)
    case = inputs.walker_case("G5", 64, 961)
    run("rslice/G5_dbl", dis.RSliceSampler,
        dict(ndim=5, ncdim=5, slices=4, nonbounded=None, periodic=None,
             reflective=None), case, 12, 5100,
        extra=dict(slice_doubling=True))
    # tiny scale forces many expansions
    run("rslice/G5_tiny", dis.RSliceSampler,
        dict(ndim=5, ncdim=5, slices=2, nonbounded=None, periodic=None,
             reflective=None), case, 8, 5200, scale=0.02)
    for pname, nw, slices in (("G5", 8, 2), ("E3", 8, 2)):
        case = inputs.walker_case(pname, 64, 970 + slices)
        d = case["problem"].ndim
        run(f"slice/{pname}", dis.SliceSampler,
            dict(ndim=d, ncdim=d, slices=slices, nonbounded=None,
                 periodic=None, reflective=None), case, nw, 6000 + slices)
    case = inputs.walker_case("G5", 64, 981)
    run("slice/G5_dbl", dis.SliceSampler,
        dict(ndim=5, ncdim=5, slices=2, nonbounded=None, periodic=None,
             reflective=None), case, 8, 6100, extra=dict(slice_doubling=True))

    # ---- unif inside a bound (internal_samplers.py:243-340) ----
    def run_unif(tag, prob, bound, nwalk, seedbase, loglstar, ndim, ncdim):
        seeds = np.random.SeedSequence(seedbase).spawn(nwalk)
        kw = dict(bound=bound, ndim=ndim, n_cluster=ncdim, nonbounded=None)
        res = []
        for i in range(nwalk):
            arg = dis.SamplerArgument(
                u=np.zeros(ndim), loglstar=loglstar, axes=None, scale=1.,
                prior_transform=prob.prior_transform,
                loglikelihood=prob.loglikelihood, rseed=seeds[i], kwargs=kw)
            res.append(dis.UniformBoundSampler.sample(arg))
        g[f"{tag}/u"] = np.array([r.u for r in res])
        g[f"{tag}/v"] = np.array([r.v for r in res])
        g[f"{tag}/logl"] = np.array([float(r.logl) for r in res])
        g[f"{tag}/ncalls"] = np.array([r.ncalls for r in res], dtype=np.int64)
        g[f"{tag}/loglstar"] = np.float64(loglstar)
        g[f"{tag}/seedbase"] = np.int64(seedbase)

    prob = inputs.problem("C1")
    pts = inputs.cloud("g3")
    single = db.Ellipsoid(3)
    single.update(pts, rstate=np.random.default_rng(5))
    lstar = float(np.quantile(
        prob.loglikelihood_many(prob.prior_transform_many(pts)), 0.3))
    run_unif("unif/C1_single", prob, single, 16, 7000, lstar, 3, 3)
    prob = inputs.problem("G5")
    pts = inputs.cloud("two5")
    multi = db.MultiEllipsoid(5)
    multi.update(pts, rstate=np.random.default_rng(5))
    multi.scale_to_logvol(multi.logvol + 5 * np.log(3.0))  # make them overlap
    lstar = float(np.quantile(
        prob.loglikelihood_many(prob.prior_transform_many(pts)), 0.3))
    run_unif("unif/G5_multi", prob, multi, 16, 7100, lstar, 5, 5)
    g["unif/G5_multi/nells"] = np.int64(multi.nells)
    # ncdim < ndim with a 3-D bound on a 5-D problem
    b3 = db.Ellipsoid(3)
    b3.update(pts[:, :3].copy(), rstate=np.random.default_rng(5))
    run_unif("unif/G5_nc3", prob, b3, 8, 7200, lstar - 30.0, 5, 3)

    # ---- unit cube (internal_samplers.py:364-441) ----
    prob = inputs.problem("C1")
    seeds = np.random.SeedSequence(7300).spawn(8)
    res = []
    for i in range(8):
        arg = dis.SamplerArgument(
            u=np.zeros(3), loglstar=-60.0, axes=None, scale=1.,
            prior_transform=prob.prior_transform,
            loglikelihood=prob.loglikelihood, rseed=seeds[i],
            kwargs=dict(ndim=3))
        res.append(dis.UnitCubeSampler.sample(arg))
    g["unitcube/C1/u"] = np.array([r.u for r in res])
    g["unitcube/C1/logl"] = np.array([float(r.logl) for r in res])
    g["unitcube/C1/ncalls"] = np.array([r.ncalls for r in res], dtype=np.int64)

    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


def gen_rng(out):
    """PCG64 / SeedSequence / ziggurat known answers from NumPy itself (the
    reference's RNG: utils.py:993-1009)."""
    g = {}
    ss = np.random.SeedSequence([11, 22, 33, 44])
    kids = ss.spawn(5)
    g["ss/entropy"] = np.array([11, 22, 33, 44], dtype=np.uint64)
    g["ss/child_state4"] = np.array(
        [k.generate_state(4, np.uint64) for k in kids])
    st = []
    for k in kids:
        s = np.random.PCG64(k).state["state"]
        st.append([s["state"] >> 64, s["state"] & (2**64 - 1),
                   s["inc"] >> 64, s["inc"] & (2**64 - 1)])
    g["ss/pcg_state"] = np.array(st, dtype=np.uint64)
    rng = np.random.Generator(np.random.PCG64(kids[2]))
    g["stream/normals"] = rng.standard_normal(5000)
    g["stream/uniforms"] = rng.random(100)
    g["stream/normals2"] = rng.standard_normal(7)
    s = rng.bit_generator.state["state"]
    g["stream/final_state"] = np.array(
        [s["state"] >> 64, s["state"] & (2**64 - 1), s["inc"] >> 64,
         s["inc"] & (2**64 - 1)], dtype=np.uint64)
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


def gen_friends(out):
    """RadFriends ('balls') / SupFriends ('cubes'): two successive updates (the
    second clusters in the metric of the first), a bootstrap update, membership
    index lists, scale_to_logvol and same-seed draws."""
    from dynesty import bounding as db
    from dynesty.utils import get_seed_sequence
    import inputs
    g = {}
    for kind, cls in (("balls", db.RadFriends), ("cubes", db.SupFriends)):
        for name in inputs.CLOUDS_FRIENDS:
            if kind == "cubes" and name == "c2s":
                # 25-D: the max-norm radius is far below the 2-norm linkage length, the second
                # update finds 600 singleton clusters and the reference divides by hsmax = 0
                continue
            pts = inputs.cloud(name)
            n, d = pts.shape
            b = cls(d)
            key = f"{kind}/{name}"
            for step in (1, 2):
                b.update(pts, rstate=np.random.default_rng(7), bootstrap=0)
                for k in ("cov", "am", "axes", "axes_inv"):
                    g[f"{key}/u{step}/{k}"] = np.array(getattr(b, k)).real
                g[f"{key}/u{step}/logvol"] = np.float64(b.logvol)
            b.ctrs = pts  # what Sampler.propose_live does (sampler.py:483-484); SupFriends.update leaves it unset
            # membership of probe points: exact integer KAT
            rng = np.random.default_rng(11)
            probes = np.concatenate([pts[rng.integers(n, size=6)] + 0.3 * rng.standard_normal((6, d)) @ b.axes,
                                     rng.uniform(size=(4, d))])
            g[f"{key}/probes"] = probes
            w = [b.within(x) for x in probes]
            g[f"{key}/within_counts"] = np.array([len(x) for x in w], dtype=np.int64)
            g[f"{key}/within_idx"] = np.concatenate(w).astype(np.int64) if sum(map(len, w)) else np.zeros(0, np.int64)
            # draws from ONE generator
            rs = np.random.default_rng(13)
            g[f"{key}/samples"] = b.samples(12, rstate=rs)
            g[f"{key}/samples_state_after"] = np.array(
                [rs.bit_generator.state["state"]["state"] >> 64, rs.bit_generator.state["state"]["state"] & (2**64 - 1),
                 rs.bit_generator.state["has_uint32"], rs.bit_generator.state["uinteger"]], dtype=np.uint64)
            rs = np.random.default_rng(14)
            xq = [b.sample(rstate=rs, return_q=True) for _ in range(8)]
            g[f"{key}/sample_q_x"] = np.array([x for x, q in xq])
            g[f"{key}/sample_q_q"] = np.array([q for x, q in xq], dtype=np.int64)
            # enlarge
            lv = b.logvol + np.log(1.25)
            b.scale_to_logvol(lv)
            for k in ("cov", "am", "axes", "axes_inv"):
                g[f"{key}/scaled/{k}"] = np.array(getattr(b, k)).real
            # bootstrap radius (3 replicas) from a fresh bound of the same kind
            b2 = cls(d)
            b2.update(pts, rstate=np.random.default_rng(7), bootstrap=0)
            b2.update(pts, rstate=np.random.default_rng(9), bootstrap=3)
            for k in ("cov", "am", "axes", "axes_inv"):
                g[f"{key}/boot/{k}"] = np.array(getattr(b2, k)).real
            g[f"{key}/boot/logvol"] = np.float64(b2.logvol)
            # no-clustering variant
            b3 = cls(d)
            b3.update(pts, rstate=np.random.default_rng(7), bootstrap=0, use_clustering=False)
            g[f"{key}/noclust/cov"] = np.array(b3.cov).real
            g[f"{key}/noclust/logvol"] = np.float64(b3.logvol)
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


def gen_runs(out):
    """Short end-to-end reference runs (static NestedSampler) for the logZ
    gate.  C1 full run; C2/C3 are too slow to regenerate casually, their
    same-seed values are recorded in SURVEY.md section 8c."""
    import dynesty
    import inputs
    g = {}
    prob = inputs.problem("C1")
    s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 3,
                              nlive=500, bound='single', sample='unif',
                              rstate=np.random.default_rng(21))
    s.run_nested(dlogz=0.01, print_progress=False, add_live=False)
    r = s.results
    g["C1/logz"] = np.float64(r.logz[-1])
    g["C1/logzerr"] = np.float64(r.logzerr[-1])
    g["C1/niter"] = np.int64(r.niter)
    g["C1/ncall"] = np.int64(np.sum(r.ncall))
    np.savez_compressed(out, **g)
    print("wrote", out, dict((k, float(v)) for k, v in g.items()))


def gen_wide(out):
    """Wide-D shapes (D > 44) from the real reference: MultiEllipsoid on two / three clusters in
    48 / 64-D, Ellipsoid on the 4000 x 200 cloud (compact: centre, axis lengths, logvol, two rows of
    the covariance), RSliceSampler in 64-D and RWalkSampler in 50-D on child streams."""
    from dynesty import bounding as db
    from dynesty import internal_samplers as dis
    import inputs
    g = {}
    for d, sizes in ((48, (2000, 2000)), (64, (2000, 1800, 2200))):
        pts = inputs.blobs(d, sizes, 0.3, d)
        m = db.bounding_ellipsoids(pts)
        g[f"multi{d}/nells"] = np.int64(m.nells)
        g[f"multi{d}/ctrs"] = np.array(m.ctrs)
        g[f"multi{d}/logvol_ells"] = np.array(m.logvol_ells)
        g[f"multi{d}/covs"] = np.array(m.covs)
    pts = inputs.cloud("g200")
    e = db.bounding_ellipsoid(pts)
    g["single200/ctr"] = e.ctr
    g["single200/logvol"] = np.float64(e.logvol)
    g["single200/axlens_sorted"] = np.sort(e.axlens)
    g["single200/cov_rows"] = e.cov[[0, 17]]
    g["single200/cov_trace"] = np.float64(np.trace(e.cov))
    for tag, cls, d, kw, seedbase in (("rslice64", dis.RSliceSampler, 64, dict(slices=3), 6400),
                                       ("rwalk50", dis.RWalkSampler, 50, dict(walks=30), 5000)):
        case = inputs.wide_walker_case(d, 6, d)
        prob = case["problem"]
        smp = cls(ndim=d, ncdim=d, nonbounded=None, periodic=None, reflective=None, **kw)
        seeds = np.random.SeedSequence(seedbase).spawn(6)
        res = []
        for i in range(6):
            arg = dis.SamplerArgument(u=case["u0"][i].copy(), loglstar=case["loglstar"], axes=case["axes"],
                                      scale=0.8, prior_transform=prob.prior_transform,
                                      loglikelihood=prob.loglikelihood, rseed=seeds[i],
                                      kwargs=dict(smp.sampler_kwargs))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res.append(cls.sample(arg))
        g[f"{tag}/u"] = np.array([r.u for r in res])
        g[f"{tag}/logl"] = np.array([float(r.logl) for r in res])
        g[f"{tag}/ncalls"] = np.array([r.ncalls for r in res], dtype=np.int64)
        for key in res[0].tuning_info:
            g[f"{tag}/ti_{key}"] = np.array([r.tuning_info[key] for r in res])
        g[f"{tag}/seedbase"] = np.int64(seedbase)
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


class _Enough(Exception):
    pass


def gen_livesets(out):
    """Live sets captured from REAL reference runs ("live sets captured from
    the oracle at fixed iterations, so shapes and distributions are real"), together with what the
    reference's own MultiEllipsoid.update made of them.  C2 (25-D, nlive 2000, multi/rwalk): bound
    updates number 1, 8 and 24 of the seed-21 run; C3 (eggbox, nlive 5000, multi/rslice): updates
    1, 6 and 16.  Uniform-in-contour shells, not Gaussian clouds: different fmax / k-means behaviour."""
    import dynesty
    from dynesty import bounding as db
    import inputs
    g = {}
    for tag, pname, nlive, sample, picks in (("C2", "C2", 2000, "rwalk", (1, 8, 24)),
                                             ("C3", "C3", 5000, "rslice", (1, 6, 16))):
        prob = inputs.problem(pname)
        taken = []
        orig = db.MultiEllipsoid.update

        def spy(self, points, *a, **kw):
            spy.n += 1
            pts = np.array(points)
            r = orig(self, points, *a, **kw)
            if spy.n in picks:
                taken.append((spy.n, pts, np.array(self.ctrs), np.array(self.covs),
                              np.array(self.logvol_ells), float(self.logvol)))
            if spy.n >= max(picks):
                raise _Enough
            return r
        spy.n = 0
        db.MultiEllipsoid.update = spy
        try:
            s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim,
                                      nlive=nlive, bound='multi', sample=sample,
                                      rstate=np.random.default_rng(21))
            try:
                s.run_nested(dlogz=0.01, print_progress=False)
            except _Enough:
                pass
        finally:
            db.MultiEllipsoid.update = orig
        for i, (n, pts, ctrs, covs, lve, lv) in enumerate(taken):
            g[f"{tag}/{i}/update_no"] = np.int64(n)
            g[f"{tag}/{i}/live_u"] = pts
            g[f"{tag}/{i}/nells"] = np.int64(len(ctrs))
            g[f"{tag}/{i}/ctrs"] = ctrs
            g[f"{tag}/{i}/covs"] = covs
            g[f"{tag}/{i}/logvol_ells"] = lve
            g[f"{tag}/{i}/logvol"] = np.float64(lv)
            print(tag, i, "update", n, "nells", len(ctrs), "logvol", lv, flush=True)
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


def gen_nsloop(out):
    """The run loop's bookkeeping from a REAL NestedSampler run (G5, nlive 200, multi/rwalk, queue of
    64 through a serial pool): for three queue fills the live log-likelihoods at the moment of the
    fill, the queue's (logl, ncalls) in order, and the dead points / evidence history the run
    recorded until the next fill; plus the whole run's dead and final live log-likelihoods with its
    final ln Z, error and information.  Pins oracle/nested_ref.py (sampler.py:741-776, 1070-1185,
    780-930; utils.py:1470-1492)."""
    import dynesty
    from dynesty import sampler as dsamp
    import inputs
    prob = inputs.problem("G5")
    nlive, K = 200, 64

    class SerialPool:
        size = K

        def map(self, f, x):
            return list(map(f, x))
    fills = []
    orig = dsamp.Sampler._fill_queue

    def spy(self, loglstar):
        snap = dict(it=len(self.saved_run['logl']), live_logl=np.array(self.live_logl),
                    loglstar_arg=float(loglstar), plateau_mode=bool(self.plateau_mode),
                    plateau_counter=int(getattr(self, "plateau_counter", 0) or 0),
                    plateau_logdvol=float(getattr(self, "plateau_logdvol", 0.) or 0.),
                    ncall=int(self.ncall))
        orig(self, loglstar)
        snap["q_logl"] = np.array([float(r.logl) for r in self.queue])
        snap["q_ncalls"] = np.array([int(r.ncalls) for r in self.queue], dtype=np.int64)
        fills.append(snap)
    dsamp.Sampler._fill_queue = spy
    # run_nested replaces the evidence history by compute_integrals' at the very end
    # (sampler.py:1342-1348): keep what the recurrence (progress_integration) had recorded
    rec = {}
    orig_ci = dsamp.compute_integrals

    def spy_ci(**kw):
        for key in ("logz", "logzvar", "h"):
            rec[key] = np.array(s.saved_run[key], dtype=np.float64)
        return orig_ci(**kw)
    dsamp.compute_integrals = spy_ci
    try:
        s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive,
                                  bound='multi', sample='rwalk', walks=20, pool=SerialPool(),
                                  queue_size=K, rstate=np.random.default_rng(77))
        s.run_nested(dlogz=0.05, print_progress=False)
    finally:
        dsamp.Sampler._fill_queue = orig
        dsamp.compute_integrals = orig_ci
    sr = dict(s.saved_run.items()) if hasattr(s.saved_run, "items") else s.saved_run
    final = {key: np.array(s.saved_run[key], dtype=np.float64) for key in ("logz", "logzvar", "h")}
    sr = {key: s.saved_run[key] for key in ("logl", "id", "logvol")}
    sr.update(rec)  # the per-iteration history below is the recurrence's
    niter = len(sr['logl']) - nlive  # add_live appended the final live points
    g = {"nlive": np.int64(nlive), "K": np.int64(K), "dlogz": np.float64(0.05),
         "nfills": np.int64(len(fills)), "niter": np.int64(niter),
         "run/logl": np.array(sr['logl'], dtype=np.float64),
         "run/logvol": np.array(sr['logvol'], dtype=np.float64),
         # the per-point bookkeeping of saved_run (sampler.py:1165-1182, 870-890)
         "run/id": np.array(s.saved_run['id'], dtype=np.int64),
         "run/it": np.array(s.saved_run['it'], dtype=np.int64),
         "run/nc": np.array(s.saved_run['nc'], dtype=np.int64),
         # every fill of the run, so that the whole loop can be replayed
         "fills/live_logl0": fills[0]["live_logl"],
         "fills/q_logl": np.array([f["q_logl"] for f in fills]),
         "fills/q_ncalls": np.array([f["q_ncalls"] for f in fills]),
         "run/ncall_init": np.int64(fills[0]["ncall"]),
         # final values of the recurrence (what the stopping rule saw) ...
         "rec/logz_final": np.float64(rec["logz"][-1]), "rec/logzvar_final": np.float64(rec["logzvar"][-1]),
         "rec/h_final": np.float64(rec["h"][-1]),
         # ... and compute_integrals' per-point history (what Results reports)
         "run/logz": final["logz"], "run/logzvar": final["logzvar"], "run/h": final["h"],
         "run/logz_final": np.float64(s.results.logz[-1]),
         "run/logzerr_final": np.float64(s.results.logzerr[-1]),
         "run/h_final": np.float64(s.results.information[-1]),
         "run/ncall": np.int64(s.ncall)}
    picks = (2, len(fills) // 2, len(fills) - 1)
    for tag, f in zip("abc", picks):
        a = fills[f]
        b = fills[f + 1]["it"] if f + 1 < len(fills) else niter
        i0 = a["it"]
        g[f"fill_{tag}/index"] = np.int64(f)
        g[f"fill_{tag}/it0"] = np.int64(i0)
        g[f"fill_{tag}/live_logl"] = a["live_logl"]
        g[f"fill_{tag}/q_logl"] = a["q_logl"]
        g[f"fill_{tag}/q_ncalls"] = a["q_ncalls"]
        # state after iteration i0 - 1 (the fill is asked for inside iteration i0)
        for key in ("logz", "logzvar", "h", "logvol", "logl"):
            g[f"fill_{tag}/state_{key}"] = np.float64(sr[key][i0 - 1] if i0 > 0 else
                                                      dict(logz=-1.e300, logzvar=0., h=0., logvol=0.,
                                                           logl=-1.e300)[key])
        g[f"fill_{tag}/dead_logl"] = np.array(sr['logl'][i0:b], dtype=np.float64)
        g[f"fill_{tag}/dead_slot"] = np.array(sr['id'][i0:b], dtype=np.int64)
        g[f"fill_{tag}/logz"] = np.array(sr['logz'][i0:b], dtype=np.float64)
        g[f"fill_{tag}/logzvar"] = np.array(sr['logzvar'][i0:b], dtype=np.float64)
        g[f"fill_{tag}/h"] = np.array(sr['h'][i0:b], dtype=np.float64)
        g[f"fill_{tag}/is_last"] = np.bool_(f + 1 == len(fills))
        g[f"fill_{tag}/state_plateau_mode"] = np.bool_(a["plateau_mode"])
        g[f"fill_{tag}/state_plateau_counter"] = np.int64(a["plateau_counter"])
        g[f"fill_{tag}/state_plateau_logdvol"] = np.float64(a["plateau_logdvol"])
        print("fill", tag, f, "it0", i0, "deaths", b - i0)
    lv = np.array(sr['logvol'][:niter], dtype=np.float64)
    ladder = np.abs(np.diff(lv, prepend=0.) + np.log((nlive + 1.) / nlive)) > 1e-12
    g["run/first_plateau_it"] = np.int64(np.argmax(ladder) if ladder.any() else niter)
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays; niter", niter, "fills", len(fills), "logz", g["run/logz_final"])


if __name__ == "__main__":
    import_reference()
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    which = sys.argv[1:] or ["bounding", "proposals", "rng", "runs", "friends", "wide", "nsloop"]
    if "bounding" in which:
        gen_bounding(os.path.join(gdir, "bounding.npz"))
    if "proposals" in which:
        gen_proposals(os.path.join(gdir, "proposals.npz"))
    if "rng" in which:
        gen_rng(os.path.join(gdir, "rng.npz"))
    if "runs" in which:
        gen_runs(os.path.join(gdir, "runs.npz"))
    if "friends" in which:
        gen_friends(os.path.join(gdir, "friends.npz"))
    if "wide" in which:
        gen_wide(os.path.join(gdir, "wide.npz"))
    if "nsloop" in which:
        gen_nsloop(os.path.join(gdir, "nsloop.npz"))
    if "livesets" in which:  # not in the default list: two partial reference runs (minutes)
        gen_livesets(os.path.join(gdir, "livesets.npz"))
