"""Solver-parameter scan on a frontier batch of config 2: SLOW count, total iterations, batch time."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 10
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=int(os.environ.get("SEED", 0)))
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-512:]
grid = [dict(rho_scale=rs, relax=rx, rho_f_ratio=rf) for rs in (2.0, 4.0, 8.0) for rx in (1.6, 1.8) for rf in (0.05, 0.1, 0.2)]
grid += [dict(rho_scale=4.0, check_every=50), dict(rho_scale=4.0, relax=1.9), dict(rho_scale=4.0, relax=1.4), dict(rho_scale=16.0)]
for kw in grid:
    Pk = omc_amd.default_params(slots=len(nodes), **kw)
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=Pk, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    print("%-62s %.2fs status %s iters median %d mean %.0f total %d" % (kw, el, st, np.median(it), it.mean(), it.sum()), flush=True)
