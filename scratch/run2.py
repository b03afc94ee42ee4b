import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
for (n, m, k, frac, kind) in [(50, 50, 1, 0.5, "noise"), (100, 100, 1, 0.2, "lowrank")][1:]:
    A, mask = make_instance(n, m, k, frac, 0, kind=kind)
    I = Inst(A, mask, 80.0, k)
    for rf, rc in [(100,100),(1000,1000),(1e4,1e4),(300,300),(3000,3000)]:
        t = time.time(); out = admm(I, rho_f=rf, rho_c=rc, iters=1000, tol=1e-9); el = time.time() - t
        h = out['hist']; fs = h[-1][1]
        it6 = next((it for it, fv, rp, rd in h if abs(fv - fs) / abs(fs) < 1e-6), None)
        print(n, kind, "rho", rf, rc, "iters", out['iters'], "f*=%.10f" % fs, "it(1e-6)=", it6, "rp %.1e rd %.1e" % (h[-1][2], h[-1][3]), "%.1fs" % el, flush=True)
