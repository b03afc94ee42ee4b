import sys; sys.path.insert(0, 'oracle')
import numpy as np, omc_oracle as orc
n, m, k = 30, 30, 1
A, mask = orc.make_instance(n, m, k, seed=0, kind="lowrank", n_indices=int(0.3 * n * m))
inst = orc.Instance(A, mask, 80.0, k)
ctype = "linear3"; dirs_all = orc.child_directions(ctype, k)
cuts = []
for d in range(3):
    r = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(rho_scale=4.0), want_certificate=False)
    x, ev = orc.breakpoint_vector(r["Y"], r["U"])
    print("depth", d, "obj", r['objective'], "its", r['iters'], "vhat", r["U"].T @ x, "dir", dirs_all[d % 4], "ev", ev)
    cuts = cuts + [(x, r["U"].copy(), dirs_all[d % len(dirs_all)])]
for q1 in [True, False]:
    r = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(rho_scale=4.0, max_iters=1500, reference_quirk_q1=q1))
    print("q1", q1, "status", r['termination_status'], "iters", r['iters'], "obj", r['objective'], "lb", r['dual_bound'])
    for h in r['hist'][::6]: print("   it %d obj %.8f lb %.8f rp %.2e rd %.2e rho %.3g" % h)
    print("   lam", r['lam'], r['rows'].kinds, "\n   res", r['residuals'])
