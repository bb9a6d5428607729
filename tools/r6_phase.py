"""Phase split of the bench-shard rebuild (timing build: `make -C dynesty_amd/csrc timing`), per kernel phase and -- for
k_split's workgroup 0 -- per level: python tools/r6_phase.py [R ...]"""
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
    out = (C.c_longlong * 16)(); lv = (C.c_longlong * 256)()
    ctx.lib.dh_rebuild_timing(out, 1); ctx.lib.dh_rebuild_timing_levels(lv, 1)
    el = (C.c_longlong * 128)(); ctx.lib.dh_rebuild_timing_ell(el, 1)
    reps = 10
    for _ in range(reps):
        s.rebuild(); ctx.sync()
    ctx.lib.dh_rebuild_timing(out, 1); ctx.lib.dh_rebuild_timing_levels(lv, 1); ctx.lib.dh_rebuild_timing_ell(el, 1)
    print("runs", R, {n: round(out[i] / reps / 100.0, 1) for i, n in enumerate(names)}, "(units of 100 clock64 ticks per rebuild, workgroup 0 of each kernel)")
    for L in range(8):
        r = [lv[L * 16 + i] / reps for i in range(16)]
        if r[4] == 0:
            continue
        print(f"  k_split level {L}: stage {r[0]/100:.0f}  lloyd {r[1]/100:.0f} ({r[4]:.1f} iterations, {r[1]/max(r[4],1)/100:.1f} each)  partition {r[2]/100:.0f}  children {r[3]/100:.0f}   parts {r[5]:.0f} points {r[6]:.0f}")
        it = max(r[4], 1)
        print("      per iteration: " + "  ".join(f"{n} {r[i]/it:.0f}" for n, i in (("vq", 8), ("ballot+count", 9), ("sums", 10), ("barrier", 11), ("update", 12), ("parts", 13), ("barrier", 14))))
    for L in range(8):
        r = [el[L * 16 + i] / reps / 100.0 for i in range(16)]
        if sum(r) == 0:
            continue
        print(f"  k_ell level {L} (workgroup 0, its nodes): " + "  ".join(f"{names[i]} {r[i]:.0f}" for i in (0, 1, 13, 14, 15, 3, 5)) + f"   (fmax includes ldl+inv, squaring, axis)")
