"""Where a pool.map call of the drop-in goes (tap B): one queue fill of 512 rwalk walkers at C2
through samplers.run_rwalk, split into its host stages and the device call.  python tools/marsh_prof.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from collections import namedtuple
from dynesty_amd import samplers, backend, bounding
SamplerArgument = namedtuple('SamplerArgument', ['u', 'loglstar', 'axes', 'scale', 'prior_transform', 'loglikelihood', 'rseed', 'kwargs'])
prob = bench.c2_problem()
K, D = 512, prob.ndim
u0, loglstar = bench.make_shard(prob, 1, 2000, 1000)
bound = bounding.HipMultiEllipsoid(D)
bound.update(u0)
rng = np.random.default_rng(3)
ss = np.random.SeedSequence(11)
kw = dict(problem=prob, walks=45)


def mk():
    seeds = ss.spawn(K)
    return [SamplerArgument(u=u0[rng.integers(2000)], loglstar=loglstar, axes=bound.get_random_axes(rng), scale=0.27,
                            prior_transform=prob.prior_transform, loglikelihood=prob.loglikelihood, rseed=seeds[i], kwargs=kw)
            for i in range(K)]


def timeit(fn, n=30):
    fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e3


al = [mk() for _ in range(40)]
it = iter(al * 10)
out = {"run_rwalk_ms": timeit(lambda: samplers.run_rwalk(next(it)))}
a = al[0]
out["start_points_ms"] = timeit(lambda: samplers._start_points(a))
out["frames_ms"] = timeit(lambda: samplers._frames(a))
out["streams_ms_incl_device_seed_hash"] = timeit(lambda: samplers._Streams([x.rseed for x in a]))
be = backend.get_backend()
st = samplers._Streams([x.rseed for x in a])
axes, idx = samplers._frames(a)
u = samplers._start_points(a)
out["device_rwalk_batch_ms"] = timeit(lambda: be.rwalk_batch(prob, u, axes, 0.27, loglstar, 45, st.states, axes_idx=idx, ncdim=D))
o = be.rwalk_batch(prob, u, axes, 0.27, loglstar, 45, st.states, axes_idx=idx, ncdim=D)
acc, rej = o["accept"].tolist(), o["reject"].tolist()
out["returns_ms"] = timeit(lambda: samplers._returns(o["u"], o["v"], o["logl"], 45, [{'accept': x, 'reject': y, 'scale': 0.27} for x, y in zip(acc, rej)],
                                                     [{'n_accept': x, 'n_reject': y} for x, y in zip(acc, rej)]))
out["get_random_axes_512_ms"] = timeit(lambda: [bound.get_random_axes(rng) for _ in range(K)])
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
# the same device call after the GPU has idled for as long as dynesty's host loop takes between two fills (~30 ms)
ts = []
for _ in range(20):
    time.sleep(0.03)
    t = time.perf_counter()
    be.rwalk_batch(prob, u, axes, 0.27, loglstar, 45, st.states, axes_idx=idx, ncdim=D)
    ts.append((time.perf_counter() - t) * 1e3)
print(json.dumps({"device_rwalk_batch_ms_after_30ms_idle": {"median": round(float(np.median(ts)), 3), "min": round(min(ts), 3), "max": round(max(ts), 3)}}))
