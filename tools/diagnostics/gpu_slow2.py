import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
from omc_amd_pkg import _lib
def probe(A, mask, k, ctype, depth, sc, label):
    eng = omc_amd.Engine(A, mask, 80.0, k)
    P = omc_amd.default_params(rho_scale=sc)
    nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, ctype, params=P)
    out = eng.matrix_completion_SDP_relaxation(nodes, ctype, params=P, want_Y=False, want_X=False)
    st = np.array([o["status_code"] for o in out]); it = np.array([o["iters"] for o in out])
    slow = [i for i in range(len(nodes)) if st[i] == 1 or it[i] > 1500]
    fast = [i for i in range(len(nodes)) if st[i] == 0 and it[i] <= 600][:6]
    sel = slow[:12] + fast
    P5 = omc_amd.default_params(rho_scale=sc, max_iters=400)
    o5 = eng.matrix_completion_SDP_relaxation([nodes[i] for i in sel], ctype, params=P5, want_Y=False, want_X=False)
    rp = np.zeros(len(sel)); rd = np.zeros(len(sel)); _lib.check(eng._lib.omc_debug_residuals(eng._h, _lib.ptr(rp), _lib.ptr(rd)))
    res = {}
    for scl in (sc / 16, sc * 16):
        oo = eng.matrix_completion_SDP_relaxation([nodes[i] for i in sel], ctype, params=P, want_Y=False, want_X=False, rho_scales=[scl] * len(sel))
        res[scl] = [o["iters"] if o["status_code"] == 0 else -o["iters"] for o in oo]
    print("====", label, "rho", eng.solver_info()["rho"])
    for t, i in enumerate(sel):
        print("%s node %3d L=%d  iters@sc %5d st %d | at 400: gap %.1e rp %.1e rd %.1e rp/rd %.2g | iters sc/16 %5d  sc*16 %5d" % ("SLOW" if i in slow else "fast", i, len(nodes[i]), it[i], st[i],
              (o5[t]["objective"] - o5[t]["dual_bound"]) / abs(o5[t]["objective"]), rp[t], rd[t], rp[t] / max(rd[t], 1e-300), res[sc / 16][t], res[sc * 16][t]))
    eng.close()
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
probe(A, mask, 1, "linear", 6, 4.0, "config2 depth6")
A, mask = omc_amd.pkg.data.readme_instance(20, 25, 1)
probe(A, mask, 1, "linear", 5, 16.0, "readme 20x25 depth5")
