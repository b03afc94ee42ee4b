import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
n, m, k, frac, kind = 20, 25, 1, 0.5, "noise"
A, mask = make_instance(n, m, k, frac, 0, kind=kind)
I = Inst(A, mask, 80.0, k)
rho = 0.5 * 80 * (A[mask] ** 2).sum() / m
print("rho", rho)
cuts = []; Y0 = U0 = None
for depth in range(8):
    t = time.time()
    o = admm(I, cuts=cuts, rho_f=rho, rho_c=rho, iters=4000, tol=1e-9, Y0=Y0, U0=U0)
    fv = o['hist'][-1][1]; lb = dual_bound(I, o)
    x, w = separation(o['Y'], o['U'])
    h = o['hist']; it6 = next((it for it, f_, rp, rd in h if abs(f_ - fv) / abs(fv) < 1e-6), None)
    print("depth", depth, "f %.8f LB %.8f gap %.1e" % (fv, lb, (fv - lb) / fv), "iters", o['iters'], "it6", it6, "eig", w[0], "v", (x @ o['U']), "|U|", np.linalg.norm(o['U']), "lam", o['lam'][-3:].round(3), "%.1fs" % (time.time() - t))
    d = "left" if depth % 2 == 0 else "right"
    cuts = cuts + [(x, o['U'].copy(), [d])]
    Y0, U0 = o['Y'], o['U']
