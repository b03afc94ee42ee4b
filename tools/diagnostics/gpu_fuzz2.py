"""Randomised parity sweep for alternating minimisation (k = 1, 2, 3) and the Shor-minor kernels against the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, omc_amd
import omc_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
t0 = time.time(); na = nab = ns = nsb = nskip = ninf = 0; worst = 0.0
while time.time() - t0 < budget:
    n = int(rng.integers(6, 20)); m = n + int(rng.integers(0, 8)); k = int(rng.choice([1, 2, 2, 3]))
    if k >= n // 2: continue
    ct = str(rng.choice(["linear", "linear2", "linear3"]))
    try:
        A, mask = orc.make_instance(n, m, k, seed=int(rng.integers(1 << 30)), kind="lowrank", n_indices=max(int(rng.uniform(0.4, 0.9) * n * m), (n + m) * k))
    except ValueError:
        continue
    inst = orc.Instance(A, mask, 80.0, k); eng = omc_amd.Engine(A, mask, 80.0, k)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), k) + 0.1 * rng.standard_normal((n, k))
    dirs = orc.child_directions(ct, k)
    cuts = []
    for _ in range(int(rng.integers(0, 3))):
        x = rng.standard_normal(n); x /= np.linalg.norm(x)
        cuts.append((x, 0.5 * rng.standard_normal((n, k)) / np.sqrt(n), list(dirs[int(rng.integers(len(dirs)))])))
    if k > 1:      # random cut sets can make model_U infeasible (no node of a real tree does): skip those
        try:
            _, _, info0 = orc.altmin_u_step(inst, orc.altmin_v_step(inst, U0), cuts, ct)
        except Exception:
            nskip += 1; eng.close(); continue
        if not (info0["kkt_residual"] <= 1e-8):
            ninf += 1; eng.close(); continue
    g = eng.alternating_minimization([U0], [cuts], ct, max_iters=30)[0]
    try:
        r = orc.alternating_minimization(inst, U0, cuts, ct, max_iters=30)
    except Exception as ex_:            # the numpy oracle gives up on a numerically singular U-step (counted, not compared)
        nskip += 1; eng.close(); continue
    na += 1
    rel = abs(g["objectives"][-1] - r["objectives"][-1]) / max(1.0, abs(r["objectives"][-1])); worst = max(worst, rel)
    if g["n_iters"] != r["n_iters"] or g["converged"] != r["converged"] or rel > 1e-7:
        np.savez(os.path.join(ROOT, "gpurun_out", "altmin_mismatch_%d.npz" % nab), A=A, mask=mask, k=k, ct=ct, U0=U0, cut_x=np.array([c_[0] for c_ in cuts]), cut_U=np.array([c_[1] for c_ in cuts]), cut_dir=np.array([[orc.DIR_CODES[d_] for d_ in c_[2]] for c_ in cuts]), gU=g["U"], gV=g["V"], gobj=np.array(g["objectives"]))
        nab += 1; print("ALTMIN MISMATCH n %d m %d k %d %s cuts %d: gpu (%d its, %.10f) oracle (%d its, %.10f)" % (n, m, k, ct, len(cuts), g["n_iters"], g["objectives"][-1], r["n_iters"], r["objectives"][-1]), flush=True)
    # Shor minors on the same mask
    cl = [int(v) for v in rng.permutation(5)[: int(rng.integers(1, 4))]]
    want = np.array(orc.shor_constraints_indexes(mask, cl), dtype=np.int64).reshape(-1, 4)
    got = eng.generate_rank1_matrix_completion_Shor_constraints_indexes(cl)
    X3 = np.round(rng.standard_normal((k, n, m)) * 3) / 3 if rng.random() < 0.5 else rng.standard_normal((k, n, m))
    ex = [tuple(int(v) for v in want[i]) for i in rng.integers(0, max(len(want), 1), 5)] if len(want) else []
    nm = int(rng.choice([1, 7, 50]))
    w2 = orc.violated_shor_minors(X3, mask, cl, ex, nm); g2 = eng.generate_violated_Shor_minors(X3, cl, ex, nm)
    ns += 1
    if not np.array_equal(want, got) or [t for _, t in w2] != [t for _, t in g2] or [s_ for s_, _ in w2] != [s_ for s_, _ in g2]:
        nsb += 1; print("SHOR MISMATCH n %d m %d k %d classes %s" % (n, m, k, cl), flush=True)
    eng.close()
print("altmin cases %d mismatches %d (worst relative final-objective difference %.1e), oracle failures skipped %d, infeasible random cut sets skipped %d; Shor cases %d mismatches %d; %.0fs" % (na, nab, worst, nskip, ninf, ns, nsb, time.time() - t0))
