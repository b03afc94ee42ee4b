import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 7, c["cut_type"], params=P)
for kw in [dict(), dict(bump_max=5), dict(bump_max=5, bump_window=4), dict(bump_max=6, bump_window=4, bump_ratio=4.0), dict(bump_max=6, bump_window=4, bump_ratio=4.0, bump_after=100), dict(bump_max=0)]:
    Pk = omc_amd.default_params(rho_scale=4.0, **kw)
    t = time.time(); out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=Pk, want_Y=False, want_X=False); el = time.time() - t
    it = np.array([o["iters"] for o in out]); st = np.array([o["status_code"] for o in out])
    g = np.array([(o["objective"] - o["dual_bound"]) / abs(o["objective"]) for o in out])
    print(kw, "status", np.bincount(st, minlength=4), "sum iters", it.sum(), "median", int(np.median(it)), "solve %.2fs" % el, "worst gap %.1e" % g.max(), flush=True)
