import sys; sys.path.insert(0, 'oracle')
import numpy as np, time
import omc_oracle as orc
for (n, m, k, kind, seed, ctype, sc) in [(20, 25, 1, "readme", 1, "linear", 16.0), (30, 30, 1, "lowrank", 0, "linear", 4.0), (16, 20, 2, "lowrank", 2, "linear2", 4.0)]:
    A, mask = orc.make_instance(n, m, k, seed=seed, kind=kind, n_indices=None if kind == "readme" else int(0.3 * n * m))
    inst = orc.Instance(A, mask, 80.0, k)
    rng = np.random.default_rng(5)
    cuts = []; dirs_all = orc.child_directions(ctype, k)
    for d in range(7):
        r0 = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(rho_scale=sc, bump=0), want_certificate=False)
        r1 = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(rho_scale=sc, bump=1), want_certificate=False)
        print(n, kind, "depth", d, "| no bump: its", r0['iters'], "st", r0['termination_status'], "gap %.1e" % ((r0['objective'] - r0['dual_bound']) / abs(r0['objective'])),
              "| bump: its", r1['iters'], "st", r1['termination_status'], "gap %.1e" % ((r1['objective'] - r1['dual_bound']) / abs(r1['objective'])), "rho x%.0f" % (r1['rho'] / r0['rho']), "obj %.7f %.7f" % (r0['objective'], r1['objective']), flush=True)
        x, ev = orc.breakpoint_vector(r1["Y"], r1["U"])
        cuts = cuts + [(x, r1["U"].copy(), dirs_all[int(rng.integers(len(dirs_all)))])]
