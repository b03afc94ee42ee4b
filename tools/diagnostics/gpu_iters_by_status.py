"""Where the ADMM iterations of a frontier go: iteration counts of the certified and of the SLOW_PROGRESS nodes (config 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 9
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0, slots=1024)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
it = np.array([o["iters"] for o in out]); st = np.array([o["status_code"] for o in out])
gap = np.array([(o["objective"] - o["dual_bound"]) / max(abs(o["objective"]), 1e-12) for o in out])
print("nodes", len(out), "total iterations", it.sum())
for code, name in ((0, "optimal"), (1, "slow"), (3, "infeasible")):
    m = st == code
    if m.any():
        print("%-10s n=%4d  share of iterations %.1f%%  iters min/median/mean/max %d/%d/%.0f/%d" % (name, m.sum(), 100.0 * it[m].sum() / it.sum(), it[m].min(), np.median(it[m]), it[m].mean(), it[m].max()))
m = st == 1
if m.any():
    print("slow nodes: iteration histogram", np.histogram(it[m], bins=[0, 400, 600, 800, 1000, 1500, 2000, 2500, 3001])[0].tolist(), "bins 0,400,600,800,1000,1500,2000,2500,3000")
    print("slow nodes: final relative gap quantiles", np.quantile(gap[m], [0.1, 0.5, 0.9]).tolist())
m = st == 0
print("optimal nodes: iteration histogram", np.histogram(it[m], bins=[0, 200, 300, 400, 500, 600, 800, 1000, 1500, 3001])[0].tolist(), "bins 0,200,300,400,500,600,800,1000,1500,3000")
