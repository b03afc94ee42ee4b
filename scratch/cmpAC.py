import sys; sys.path.insert(0, 'oracle'); sys.path.insert(0, 'scratch')
import numpy as np, time
import omc_oracle as orc
from formC import relaxC
def first(hist, tol):
    for (it, obj, lb, rp, rd) in hist:
        if obj - lb <= tol * max(1, abs(obj)) and rp <= 1e-7 * 5: return it
    return None
for (n, m, k, kind, seed, ctype) in [(20, 25, 1, "readme", 1, "linear"), (30, 30, 1, "lowrank", 0, "linear"), (16, 20, 2, "lowrank", 2, "linear2")]:
    A, mask = orc.make_instance(n, m, k, seed=seed, kind=kind, n_indices=None if kind == "readme" else int(0.3 * n * m))
    inst = orc.Instance(A, mask, 80.0, k)
    rng = np.random.default_rng(5)
    cuts = []; nodes = [[]]; dirs_all = orc.child_directions(ctype, k)
    for d in range(7):
        r = relaxC(inst, cuts, ctype, params=orc.RelaxParams(max_iters=600))
        x, ev = orc.breakpoint_vector(r["Y"], r["U"])
        cuts = cuts + [(x, r["U"].copy(), dirs_all[int(rng.integers(len(dirs_all)))])]
        nodes.append(list(cuts))
    for b, c in enumerate(nodes):
        p = orc.RelaxParams(max_iters=2000, eps_gap=1e-7)
        t = time.time(); ra = orc.sdp_relaxation(inst, c, ctype, params=p, want_certificate=False); ta = time.time() - t
        t = time.time(); rc = relaxC(inst, c, ctype, params=p); tc = time.time() - t
        print(n, k, ctype, "node", b, "| A obj %.8f lb %.8f it5 %s it6 %s | C obj %.8f lb %.8f it5 %s it6 %s" % (
            ra['objective'], ra['dual_bound'], first(ra['hist'], 1e-5), first(ra['hist'], 1e-6),
            rc['objective'], rc['dual_bound'], first(rc['hist'], 1e-5), first(rc['hist'], 1e-6)), flush=True)
