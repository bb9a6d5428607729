"""Phase split (clock64 of workgroup 0 of every kernel, `make -C dynesty_amd/csrc timing`) of the bench-shard rebuild at
R runs: python tools/rb_phase_batch.py [R ...]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DYNHIP_LIB", os.path.join(ROOT, "dynesty_amd", "libdynhip_timing.so"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
ctx = _lib.Context(0)
names = ["mean", "cov", "regularize", "fmax", "kmeans(all)", "fmax:stage", "jacobi", "sort", "copy", "am", "km:vq", "km:sums",
         "km:update", "ldl+inv", "squaring", "axis"]
for R in [int(x) for x in sys.argv[1:]] or [1, 64]:
    s = bench.Shard(ctx, bench.c2_problem(), runs=R, seed=1000)
    for _ in range(3):
        s.rebuild()
    ctx.sync()
    out = (C.c_longlong * 16)()
    ctx.lib.dh_rebuild_timing(out, 1)
    reps = 10
    for _ in range(reps):
        s.rebuild()
    ctx.sync()
    ctx.lib.dh_rebuild_timing(out, 1)
    print("runs", R, {n: round(out[i] / reps / 100.0, 1) for i, n in enumerate(names)}, "(us per rebuild, workgroup 0 of each kernel)")
