import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
n, m, k, frac, kind = 50, 50, 1, 0.5, "noise"
A, mask = make_instance(n, m, k, frac, 0, kind=kind)
I = Inst(A, mask, 80.0, k)
out = admm(I, rho_f=1000, rho_c=1000, iters=600, tol=1e-9)
fv = out['hist'][-1][1]
print("root f", fv, "iters", out['iters'], "LB", dual_bound(I, out), "lam", out['lam'])
x, w = separation(out['Y'], out['U'])
print("eig", w)
for d in ["left", "right"]:
    cuts = [(x, out['U'].copy(), [d])]
    for warm in [False, True]:
        o2 = admm(I, cuts=cuts, rho_f=1000, rho_c=1000, iters=1500, tol=1e-9, Y0=out['Y'] if warm else None, U0=out['U'] if warm else None)
        h = o2['hist']; fs = h[-1][1]
        it6 = next((it for it, f_, rp, rd in h if abs(f_ - fs) / abs(fs) < 1e-6), None)
        print(d, "warm" if warm else "cold", "f", fs, "iters", o2['iters'], "it6", it6, "LB", dual_bound(I, o2), "lam", o2['lam'].round(3))
