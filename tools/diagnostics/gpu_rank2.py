"""Rank-2 smoke at 100 x 100 (20 % observed, linear cuts: 4 children per node): frontier throughput and statuses; altmin k = 2 timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask = omc_amd.pkg.data.generate_matrix_completion_data(2, 100, 100, 2000, seed=0)
eng = omc_amd.Engine(A, mask, 80.0, 2)
rs, log = omc_amd.pkg.bnb.autotune_rho_scale(eng, "linear"); print("autotune", rs, log, flush=True)
P = omc_amd.default_params(rho_scale=rs)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 4, "linear", params=P); print("frontier", len(nodes), flush=True)
t0 = time.perf_counter()
out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=omc_amd.default_params(rho_scale=rs, slots=len(nodes)), want_Y=True, want_X=False)
el = time.perf_counter() - t0
it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
print("rank 2, 100x100: %d nodes in %.2fs = %.1f/s; status %s; iters median %d; kernel ms %s" % (len(nodes), el, len(nodes) / el, st, np.median(it), {k: round(v["ms"]) for k, v in eng.kernel_stats().items()}), flush=True)
Ur = eng.round_Y([o["Y"] for o in out[:64]])
t0 = time.perf_counter(); am = eng.alternating_minimization(Ur, nodes[:64], "linear"); el = time.perf_counter() - t0
print("altmin k=2: 64 problems in %.3fs; converged %d; iterations median %d; best objective %.4f" % (el, sum(a["converged"] for a in am), np.median([a["n_iters"] for a in am]), min(a["objectives"][-1] for a in am)))
