"""First difference between a run of dh_ns_ensemble and its host mirror (tests/resident_mirror.py):
python tools/mirror_diag.py K bound forced [run]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from resident_mirror import mirror_run
from dynesty_amd import problems, _lib
ctx = _lib.Context(0)
K, bound, forced = int(sys.argv[1]), sys.argv[2], sys.argv[3]
run = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prob = problems.gauss_corr(13, 0.3, 5.0, "corr13")
nlive, walks, dlogz, ent = 100, 20, 0.5, [5, K, 7]
r = ctx.ns_ensemble(prob, 3, nlive, K, walks=walks, bound=bound, dlogz=dlogz, entropy=ent, rebuild_every=1,
                    want_samples=True, want_dead_logl=True, forced_exact=forced == "exact", max_iter=20000)
m = mirror_run(ctx, prob, nlive, K, walks, bound, ent, run, dlogz, forced=forced)
n = int(r["niter"][run])
a, b = r["dead_logl"][run, :n], np.array(m["dead_logl"])
k = min(len(a), len(b))
d = np.nonzero(np.abs(a[:k] - b[:k]) > 1e-12 * np.abs(a[:k]))[0]
print("n", n, len(b), "first value diffs", d[:5], "slot diffs", np.nonzero(r["dead_id"][run, :k] != np.array(m["dead_slot"])[:k])[0][:5])
if len(d):
    i = int(d[0])
    print("death", i, "fill", m["fill_of_death"][i], "slot", r["dead_id"][run, i], a[i], b[i], "src", m["dead_src"][i])
    # when did that slot get its value: the last earlier death of the same slot
    prev = [e for e in range(i) if r["dead_id"][run, e] == r["dead_id"][run, i]]
    print("slot's previous death", prev[-1:] , "its fill", [m["fill_of_death"][e] for e in prev[-1:]], "src", [m["dead_src"][e] for e in prev[-1:]])
    print("forced fills near", [f for f in m["forced_fills"] if abs(f - m["fill_of_death"][prev[-1]] if prev else 0) < 3])
