import sys; sys.path.insert(0, 'oracle')
import numpy as np, time
import omc_oracle as orc
for (n, m, k, kind, seed, ctype, scales) in [(20, 25, 1, "readme", 1, "linear", [1, 4, 16]), (30, 30, 1, "lowrank", 0, "linear", [0.25, 1, 4]), (16, 20, 2, "lowrank", 2, "linear2", [0.25, 1, 4, 16])]:
    A, mask = orc.make_instance(n, m, k, seed=seed, kind=kind, n_indices=None if kind == "readme" else int(0.3 * n * m))
    inst = orc.Instance(A, mask, 80.0, k)
    rng = np.random.default_rng(5)
    # autotune at root
    best = None
    for sc in scales:
        r = orc.sdp_relaxation(inst, params=orc.RelaxParams(max_iters=3000, rho_scale=sc), want_certificate=False)
        print(n, k, kind, "root scale", sc, "iters", r['iters'], "status", r['termination_status'], "obj %.8f lb %.8f" % (r['objective'], r['dual_bound']), flush=True)
        if r['termination_status'] == 0 and (best is None or r['iters'] < best[1]): best = (sc, r['iters'])
    sc = best[0]
    cuts = []; dirs_all = orc.child_directions(ctype, k)
    for d in range(8):
        t = time.time(); r = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(max_iters=3000, rho_scale=sc)); el = time.time() - t
        print("   depth", d, "scale", sc, "iters", r['iters'], "st", r['termination_status'], "obj %.8f lb %.8f gap %.1e res %.1e r=%d %.1fs" % (r['objective'], r['dual_bound'], (r['objective'] - r['dual_bound']) / abs(r['objective']), r['residuals']['max'], r['Q'].shape[1], el), flush=True)
        if r['termination_status'] == 3: break
        x, ev = orc.breakpoint_vector(r["Y"], r["U"])
        cuts = cuts + [(x, r["U"].copy(), dirs_all[int(rng.integers(len(dirs_all)))])]
