"""Nodes flagged INFEASIBLE with Anderson acceleration on: what does the plain iteration say about the same nodes?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 11, c["cut_type"], params=P)
half = 0.5 * float((A[mask] ** 2).sum())
plain = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=omc_amd.default_params(rho_scale=4.0, slots=len(nodes)), want_Y=False, want_X=False)
acc = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=omc_amd.default_params(rho_scale=4.0, slots=len(nodes), accel=1), want_Y=False, want_X=False)
bad = 0
for i, (p_, a_) in enumerate(zip(plain, acc)):
    if a_["status_code"] == 3 or p_["status_code"] == 3:
        print("node", i, "plain:", p_["termination_status"], "obj %.6f lb %.6f" % (p_["objective"], p_["dual_bound"]), "| accel:", a_["termination_status"], "lb %.6f" % a_["dual_bound"], "| 1/2||A||^2 = %.6f" % half)
    if a_["status_code"] != 3 and p_["status_code"] == 0 and a_["dual_bound"] > p_["objective"] * (1 + 2e-6):
        bad += 1; print("  !! accel bound above the certified plain optimum at node", i, a_["dual_bound"], p_["objective"])
lbv = np.array([a_["dual_bound"] for a_ in acc]); ob = np.array([p_["objective"] for p_ in plain]); stp = np.array([p_["status_code"] for p_ in plain])
print("nodes", len(nodes), "accel bounds above a certified plain optimum:", bad, "; max (lb_accel - obj_plain)/obj over plain-OPTIMAL nodes: %.2e" % np.max(((lbv - ob) / np.abs(ob))[(stp == 0) & np.isfinite(lbv) & (lbv > -1e200)]))
