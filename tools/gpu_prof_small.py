"""Small fixed workload for profiling: config-2 instance, 8 depth-3 nodes, 150 iterations cap."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(int(os.environ.get("CFG", 2)), seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0, max_iters=int(os.environ.get("ITERS", 150)))
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, int(os.environ.get("DEPTH", 3)), c["cut_type"], params=P)
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
print(len(nodes), [o["iters"] for o in out][:8], eng.kernel_stats(), eng.solver_info())
