"""First GPU bring-up: relaxation parity vs the oracle on small instances (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import omc_amd
import omc_oracle as orc

def run(n, m, k, kind, seed, depth, ctype="linear", gamma=80.0, rho_scale=1.0):
    A, mask = orc.make_instance(n, m, k, seed=seed, kind=kind, n_indices=None if kind == "readme" else int(0.3 * n * m))
    inst = orc.Instance(A, mask, gamma, k)
    eng = omc_amd.Engine(A, mask, gamma, k)
    # build a path of cuts with the oracle
    cuts = []; nodes = [[]]
    dirs_all = orc.child_directions(ctype, k)
    for d in range(depth):
        r = orc.sdp_relaxation(inst, cuts, ctype, params=orc.RelaxParams(rho_scale=rho_scale), want_certificate=False)
        x, ev = orc.breakpoint_vector(r["Y"], r["U"])
        cuts = cuts + [(x, r["U"].copy(), dirs_all[d % len(dirs_all)])]
        nodes.append(list(cuts))
    t0 = time.time(); ref = [orc.sdp_relaxation(inst, c, ctype, params=orc.RelaxParams(rho_scale=rho_scale), want_certificate=False) for c in nodes]; tc = time.time() - t0
    t0 = time.time(); out = eng.matrix_completion_SDP_relaxation(nodes, ctype, params=omc_amd.default_params(rho_scale=rho_scale)); tg = time.time() - t0
    print(f"--- n={n} m={m} k={k} {kind} {ctype} nodes={len(nodes)} cpu {tc:.2f}s gpu {tg:.2f}s")
    worst = 0
    for b, (g, r) in enumerate(zip(out, ref)):
        rel = abs(g["objective"] - r["objective"]) / abs(r["objective"])
        worst = max(worst, rel)
        print(f"  node {b}: gpu obj {g['objective']:.9f} lb {g['dual_bound']:.9f} it {g['iters']} {g['termination_status']} | "
              f"cpu obj {r['objective']:.9f} lb {r['dual_bound']:.9f} it {r['iters']} | rel {rel:.2e} "
              f"lmin gpu {g['lambda_min'][0]:.6f} cpu {np.linalg.eigvalsh(r['U']@r['U'].T-r['Y'])[0]:.6f} |dU| {np.abs(g['U']-r['U']).max():.1e}")
    print("  kernel stats", eng.kernel_stats())
    Xs = np.stack([o["X"] for o in out])
    ev = eng.evaluate_objective(Xs)
    ev_ref = [orc.evaluate_objective(x, A, mask, gamma) for x in Xs]
    print("  evaluate_objective max rel", np.max(np.abs(ev - ev_ref) / np.abs(ev_ref)))
    eng.close()
    return worst

if __name__ == "__main__":
    w = []
    w.append(run(12, 15, 1, "readme", 0, 2, rho_scale=8.0))
    w.append(run(20, 25, 1, "readme", 1, 5, rho_scale=16.0))
    w.append(run(16, 20, 2, "lowrank", 2, 4, "linear2", rho_scale=4.0))
    w.append(run(30, 30, 1, "lowrank", 0, 4, "linear3", rho_scale=4.0))
    w.append(run(50, 50, 1, "readme", 0, 3, rho_scale=8.0))
    print("WORST", max(w))
