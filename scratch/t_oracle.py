import sys, time; sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from omc_oracle import *
# KAT-2 fully observed root
rng = np.random.default_rng(0); n, m, k, g = 12, 15, 2, 80.0
A = rng.standard_normal((n, m)); mask = np.ones((n, m), bool)
inst = Instance(A, mask, g, k)
t = time.time(); r = sdp_relaxation(inst); print("KAT2 obj", r['objective'], "lb", r['dual_bound'], "iters", r['iters'], r['residuals'], "%.2fs" % (time.time() - t))
print(" ref formula", r['objective_reference_formula'])
# config-1 like small with cuts
A, mask = make_instance(30, 30, 1, kind="readme", seed=1)
inst = Instance(A, mask, 80.0, 1)
r = sdp_relaxation(inst); print("root", r['objective'], r['dual_bound'], r['iters'], r['residuals']['max'], r['termination_status'])
x, ev = breakpoint_vector(r['Y'], r['U']); print("ev", ev)
cuts = [(x, r['U'], ["left"])]
r2 = sdp_relaxation(inst, cuts, "linear"); print("child", r2['objective'], r2['dual_bound'], r2['iters'], r2['residuals']['max'])
# altmin
U0 = svd_rounding(np.where(mask, A, 0.0), 1)
t = time.time(); am = alternating_minimization(inst, U0); print("altmin", am['converged'], am['n_iters'], am['objectives'][-3:], "%.2fs" % (time.time() - t))
X = am['U'] @ am['V']; print("eval obj", evaluate_objective(X, A, mask, 80.0), "norm U", np.linalg.norm(am['U']))
