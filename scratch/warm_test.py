import sys; sys.path.insert(0, 'oracle')
import numpy as np, time
import omc_oracle as orc
for (n, m, k, kind, seed, ctype, sc) in [(30, 30, 1, "lowrank", 0, "linear", 4.0), (20, 25, 1, "readme", 1, "linear", 16.0)]:
    A, mask = orc.make_instance(n, m, k, seed=seed, kind=kind, n_indices=None if kind == "readme" else int(0.3 * n * m))
    inst = orc.Instance(A, mask, 80.0, k)
    rng = np.random.default_rng(5)
    cuts = []; dirs_all = orc.child_directions(ctype, k); warm = None
    for d in range(7):
        P = orc.RelaxParams(rho_scale=sc, max_iters=3000)
        rc = orc.sdp_relaxation(inst, cuts, ctype, params=P, want_certificate=False)
        rw = orc.sdp_relaxation(inst, cuts, ctype, params=P, want_certificate=False, warm=warm) if warm is not None else rc
        print(n, kind, "depth", d, "cold its", rc['iters'], "st", rc['termination_status'], "| warm its", rw['iters'], "st", rw['termination_status'], "obj %.8f %.8f" % (rc['objective'], rw['objective']), flush=True)
        warm = rw['warm']
        x, ev = orc.breakpoint_vector(rw["Y"], rw["U"])
        cuts = cuts + [(x, rw["U"].copy(), dirs_all[int(rng.integers(len(dirs_all)))])]
