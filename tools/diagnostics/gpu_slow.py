import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 6, c["cut_type"], params=P)
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
g = np.array([(o["objective"] - o["dual_bound"]) / abs(o["objective"]) for o in out]); it = np.array([o["iters"] for o in out]); st = np.array([o["status_code"] for o in out])
print("B", len(nodes), "status", np.bincount(st, minlength=4))
print("gap of SLOW nodes (sorted):", np.sort(g[st == 1]))
print("iters of SLOW:", np.sort(it[st == 1]))
print("iters of OPTIMAL: median", np.median(it[st == 0]), "max", it[st == 0].max())
for e in (1e-5, 3e-6):
    P2 = omc_amd.default_params(rho_scale=4.0, eps_gap=e)
    o2 = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P2, want_Y=False, want_X=False)
    it2 = np.array([o["iters"] for o in o2]); st2 = np.array([o["status_code"] for o in o2])
    print("eps_gap", e, "status", np.bincount(st2, minlength=4), "iters median", np.median(it2), "max", it2.max(), "solve s", eng.solver_info()["solve_seconds"])
print("eps 1e-6 solve s (last)", )
slow = [nodes[i] for i in np.flatnonzero(st == 1)]
for sc in (0.5, 1.0, 2.0, 8.0, 16.0, 64.0):
    o3 = eng.matrix_completion_SDP_relaxation(slow, c["cut_type"], params=P, want_Y=False, want_X=False, rho_scales=[sc] * len(slow))
    it3 = np.array([o["iters"] for o in o3]); st3 = np.array([o["status_code"] for o in o3]); g3 = np.array([(o["objective"] - o["dual_bound"]) / abs(o["objective"]) for o in o3])
    print("slow nodes with rho_scale", sc, "status", np.bincount(st3, minlength=4), "iters", np.sort(it3)[:10], "median gap %.1e" % np.median(g3))
