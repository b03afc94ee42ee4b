"""SLOW nodes: final penalty (relative to the initial one), residuals and gap -- is the penalty over-bumped?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
from omc_amd_pkg import _lib
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-256:]
for cap in (1000, 3000):
    Pc = omc_amd.default_params(rho_scale=4.0, max_iters=cap, slots=len(nodes))
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=Pc, want_Y=False, want_X=False)
    rp = np.zeros(len(nodes)); rd = np.zeros(len(nodes)); _lib.check(eng._lib.omc_debug_residuals(eng._h, _lib.ptr(rp), _lib.ptr(rd)))
    print("keys", sorted(out[0].keys()))
    rho0 = eng.solver_info()["rho"]
    print("cap", cap, "rho0", rho0)
    for i, o in enumerate(out):
        if o["status_code"] == 1:
            print("node %3d L=%2d iters %4d gap %.1e rho/rho0 %8.1f rp %.2e rd %.2e rp/rd %.2g" % (i, len(nodes[i]), o["iters"], (o["objective"] - o["dual_bound"]) / abs(o["objective"]), o.get("rho", float("nan")) / rho0, rp[i], rd[i], rp[i] / max(rd[i], 1e-300)))
