"""Anatomy of SLOW vs OPTIMAL nodes: per cut the interval of v = U'x, its value, x'Yx and the cut-row slack; top eigenvalues of Y."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, omc_amd
import omc_oracle as orc            # diagnostics only (cut_piece)
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=int(os.environ.get("SEED", 0)))
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 9, c["cut_type"], params=P)
nodes = nodes[-256:]
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=omc_amd.default_params(rho_scale=4.0, slots=len(nodes)), want_X=False)
slow = [i for i, o in enumerate(out) if o["status_code"] == 1][:5]
fast = [i for i, o in enumerate(out) if o["status_code"] == 0][:4]
for tag, sel in (("SLOW", slow), ("OPT ", fast)):
    for i in sel:
        o = out[i]; Y = o["Y"]; U = o["U"]
        ev = np.linalg.eigvalsh(Y)[::-1][:4]
        print("%s node %3d iters %4d gap %.1e  eig(Y) %s trace %.4f" % (tag, i, o["iters"], (o["objective"] - o["dual_bound"]) / abs(o["objective"]), np.round(ev, 5), np.trace(Y)))
        for (x, Uh, dirs) in nodes[i]:
            vhat = float(Uh[:, 0] @ x)
            lo, hi, sl, ic = orc.cut_piece(c["cut_type"], dirs[0], vhat)
            v = float(U[:, 0] @ x); q = float(x @ Y @ x)
            print("      dir %-5s vhat %+.4f  v in [%+.4f, %+.4f] v = %+.6f (to lo %.1e, to hi %.1e)  x'Yx = %.6f  g(v) = %.6f  slack %.1e" % (dirs[0], vhat, lo, hi, v, v - lo, hi - v, q, sl * v + ic, sl * v + ic - q))
