import sys; sys.path.insert(0, 'oracle')
import numpy as np, omc_oracle as orc
n, m, k = 20, 25, 1
A, mask = orc.make_instance(n, m, k, seed=1, kind="readme")
inst = orc.Instance(A, mask, 80.0, k)
cuts = []
dirs_all = orc.child_directions("linear", k)
for d in range(2):
    r = orc.sdp_relaxation(inst, cuts, "linear", want_certificate=False)
    x, ev = orc.breakpoint_vector(r["Y"], r["U"])
    cuts = cuts + [(x, r["U"].copy(), dirs_all[d % 2])]
p = orc.RelaxParams(max_iters=3000)
r = orc.sdp_relaxation(inst, cuts, "linear", params=p)
for h in r['hist'][::8]: print("it %d obj %.9f lb %.9f gap %.2e rp %.2e rd %.2e" % (h[0], h[1], h[2], (h[1]-h[2])/h[1], h[3], h[4]))
print("lam", r['lam'], [ (kd) for kd in r['rows'].kinds])
print(r['residuals'])
print("---- variations")
for kw in [dict(rho_f_ratio=1.0, relax=1.0), dict(rho_f_ratio=1.0, relax=1.6), dict(rho_f_ratio=0.1, relax=1.0), dict(rho_f_ratio=0.3, relax=1.6), dict(rho_scale=0.3), dict(rho_scale=3.0), dict(rho_scale=0.1, rho_f_ratio=1.0)]:
    p = orc.RelaxParams(max_iters=3000, **kw)
    r = orc.sdp_relaxation(inst, cuts, "linear", params=p, want_certificate=False)
    h = r['hist'][-1]
    print(kw, "iters", r['iters'], "obj %.9f lb %.9f gap %.2e rp %.2e rd %.2e" % (h[1], h[2], (h[1]-h[2])/h[1], h[3], h[4]))
