"""Do the SLOW nodes converge with another initial penalty?  (portfolio / restart idea)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-256:]
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
slow = [i for i, o in enumerate(out) if o["status_code"] == 1]
print("slow nodes", len(slow))
sel = [nodes[i] for i in slow]
table = {}
for sc in (0.25, 1.0, 16.0, 64.0, 256.0):
    for bump in (6, 0):
        o2 = eng.matrix_completion_SDP_relaxation(sel, c["cut_type"], params=omc_amd.default_params(rho_scale=sc, bump_max=bump), want_Y=False, want_X=False)
        table[(sc, bump)] = [(o["status_code"], o["iters"], (o["objective"] - o["dual_bound"]) / abs(o["objective"])) for o in o2]
        print("rho_scale %6.2f bump_max %d: optimal %2d / %d ; iters of the optimal ones: %s" % (sc, bump, sum(1 for t in table[(sc, bump)] if t[0] == 0), len(sel), [t[1] for t in table[(sc, bump)] if t[0] == 0][:12]), flush=True)
best = [min((t[i][1] if t[i][0] == 0 else 10 ** 9) for t in table.values()) for i in range(len(sel))]
print("nodes that converge for at least one setting:", sum(1 for b_ in best if b_ < 10 ** 9), "of", len(sel), "best iters", [b_ if b_ < 10 ** 9 else -1 for b_ in best])
