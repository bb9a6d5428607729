"""Per-workgroup wall-clock stamps of the level kernels of ONE bench-shard rebuild (timing build): when each workgroup of
k_split / k_ell started and ended, relative to the kernel's first start.  python tools/r6_wgclock.py R"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DYNHIP_LIB", os.path.join(ROOT, "dynesty_amd", "libdynhip_timing.so"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
ctx = _lib.Context(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = bench.Shard(ctx, bench.c2_problem(), runs=R, seed=1000)
for _ in range(5):
    s.rebuild()
ctx.sync()
N = 4096
buf = (C.c_longlong * (2 * N))()
for kern, name in ((0, "k_split"), (1, "k_ell")):
    for L in range(6):
        ctx.lib.dh_rebuild_wg_clock(buf, kern, L)
        a = np.frombuffer(buf, dtype=np.int64).reshape(N, 2).copy()
        ok = (a[:, 0] > 0) & (a[:, 1] >= a[:, 0])
        if not ok.any():
            continue
        a = a[ok]
        t0 = a[:, 0].min()
        st = (a[:, 0] - t0) / 100.0
        du = (a[:, 1] - a[:, 0]) / 100.0
        en = (a[:, 1] - t0) / 100.0
        busy = du > 2.0
        q = lambda x, p: float(np.percentile(x, p)) if len(x) else float('nan')
        print(f"{name} level {L}: {len(a)} workgroups, {int(busy.sum())} busy (> 2 us); kernel span {en.max():.1f} us; "
              f"busy starts p50 {q(st[busy],50):.1f} p90 {q(st[busy],90):.1f} max {q(st[busy],100):.1f}; "
              f"busy durations p10 {q(du[busy],10):.1f} p50 {q(du[busy],50):.1f} p90 {q(du[busy],90):.1f} max {q(du[busy],100):.1f}; "
              f"idle durations p50 {q(du[~busy],50):.2f} max {q(du[~busy],100):.2f}, idle starts max {q(st[~busy],100):.1f}")
