"""Node counts of the bench-shard rebuild, repeated: a tree that changes between repetitions (or between builds) is a race.
python tools/r6_nodes.py [R] [reps]   (DYNHIP_LIB selects the library)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _lib.Context(0)
lib, h = ctx.lib, ctx.handle
s = bench.Shard(ctx, bench.c2_problem(), runs=R, seed=1000)
d_nn = ctx.malloc(R * 4)
seen = {}
for i in range(reps):
    ctx._check(lib.dh_rebuild_batch_dev(h, s.runs, s.d_u0, s.nlive, s.d, 0, s.me, s.d_nells, s.d_status, s.d_ctrs, s.d_covs,
                                        s.d_ams, s.d_axes, s.d_axl, s.d_lv, None, d_nn))
    ctx.sync()
    nn = tuple(ctx.from_device(d_nn, (R,), np.int32).tolist())
    st = tuple(ctx.from_device(s.d_status, (R,), np.int32).tolist())
    lv = ctx.from_device(s.d_lv, (R, s.me), np.float64)[:, 0].tobytes()
    seen[(nn, st, lv)] = seen.get((nn, st, lv), 0) + 1
for (nn, st, _), c in seen.items():
    print(f"{c:3d} x nodes {nn[:8]} status {st[:8]}")
print("distinct outcomes:", len(seen))
