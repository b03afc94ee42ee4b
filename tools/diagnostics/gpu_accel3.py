import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
from omc_amd_pkg import _lib
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P0 = omc_amd.default_params(rho_scale=4.0, accel=0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 8, c["cut_type"], params=P0)
nodes = nodes[-256:]
for kw in (dict(accel=1), dict(accel=1, bump_max=0), dict(accel=1, aa_safeguard=2.0), dict(accel=1, aa_reg=1e-6)):
    P = omc_amd.default_params(rho_scale=4.0, slots=len(nodes), **kw)
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
    acc = np.zeros(len(nodes), np.int32); rej = np.zeros(len(nodes), np.int32); _lib.check(eng._lib.omc_debug_aa(eng._h, _lib.ptr(acc), _lib.ptr(rej)))
    rp = np.zeros(len(nodes)); rd = np.zeros(len(nodes)); _lib.check(eng._lib.omc_debug_residuals(eng._h, _lib.ptr(rp), _lib.ptr(rd)))
    it = np.array([o["iters"] for o in out]); st = np.array([o["status_code"] for o in out])
    print(kw, "status", np.bincount(st, minlength=4), "iters total", it.sum(), "accepted total", acc.sum(), "rejected total", rej.sum())
    for i in np.where(st == 1)[0][:30]:
        o = out[i]
        print("   node %3d iters %4d gap %.1e acc %3d rej %3d rp %.1e rd %.1e" % (i, o["iters"], (o["objective"] - o["dual_bound"]) / abs(o["objective"]), acc[i], rej[i], rp[i], rd[i]))
