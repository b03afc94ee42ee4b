"""Penalty-bump limits on a full frontier batch: SLOW count and total iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-512:]
for kw in (dict(), dict(bump_after=50), dict(bump_after=200), dict(bump_ratio=3.0), dict(bump_ratio=6.0), dict(bump_window=2), dict(bump_max=3, bump_factor=3.0), dict(bump_max=2, bump_factor=3.0), dict(bump_max=2, bump_factor=5.0)):
    Pk = omc_amd.default_params(rho_scale=4.0, slots=len(nodes), **kw)
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=Pk, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    gaps = np.array([(o["objective"] - o["dual_bound"]) / abs(o["objective"]) for o in out])
    print("%-44s %.2fs status %s iters median %d mean %.0f total %d  worst gap %.1e" % (kw, el, st, np.median(it), it.mean(), it.sum(), gaps.max()), flush=True)
