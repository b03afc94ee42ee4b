"""What do the SLOW_PROGRESS nodes gain from their last 2000 iterations?  Relative gap (objective - dual bound) of the
nodes that end SLOW at max_iters = 3000, at caps 500 / 1000 / 2000 / 3000, plus the spread of the bounds in the batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-256:]
res = {}
for cap in (500, 1000, 2000, 3000):
    Pc = omc_amd.default_params(rho_scale=4.0, max_iters=cap, slots=len(nodes))
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=Pc, want_Y=False, want_X=False)
    res[cap] = (np.array([o["objective"] for o in out]), np.array([o["dual_bound"] for o in out]), np.array([o["status_code"] for o in out]), np.array([o["iters"] for o in out]))
obj, lb, st, it = res[3000]
slow = np.where(st == 1)[0]
print("nodes", len(nodes), "SLOW", len(slow), "objective range of the batch %.6f .. %.6f" % (obj.min(), obj.max()))
best = obj.copy()
print("node   iters |  rel gap (obj-lb)/obj at cap 500 / 1000 / 2000 / 3000 |  lb@500 lb@1000 lb@2000 lb@3000 relative to final obj")
for i in slow:
    g = [(res[cap][0][i] - res[cap][1][i]) / abs(res[cap][0][i]) for cap in (500, 1000, 2000, 3000)]
    l = [res[cap][1][i] / obj[i] for cap in (500, 1000, 2000, 3000)]
    print("%4d  %5d | %.1e %.1e %.1e %.1e | %.6f %.6f %.6f %.6f  st@caps %s" % (i, it[i], *g, *l, [int(res[cap][2][i]) for cap in (500, 1000, 2000, 3000)]))
