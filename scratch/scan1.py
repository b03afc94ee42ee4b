import sys; sys.path.insert(0, '/root/repo/scratch')
from proto2 import *
n, m, k, frac, kind = 100, 100, 1, 0.2, "lowrank"
A, mask = make_instance(n, m, k, frac, 0, kind=kind)
I = Inst(A, mask, 80.0, k)
rho0 = 0.5 * 80 * (A[mask] ** 2).sum() / m
fs = 49.41861188
for rr in [0.03, 0.1, 0.3]:
    for rs in [0.3, 1.0, 3.0, 10.0]:
        t = time.time(); o = admm2(I, iters=1200, tol=1e-9, relax=1.6, rf_ratio=rr, rho=rho0 * rs); el = time.time() - t
        h = o['hist']
        def itacc(e): return next((it for i_, (it, f_, rp, rd, r_) in enumerate(h) if all(abs(x[1] - fs) / abs(fs) < e for x in h[i_:])), None)
        print("rf_ratio", rr, "rho_scale", rs, "iters", o['iters'], "it1e-5", itacc(1e-5), "it1e-6", itacc(1e-6), "f", h[-1][1], "%.1fs" % el, flush=True)
