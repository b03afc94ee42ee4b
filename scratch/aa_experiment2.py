"""AA variant that the GPU can mirror: Gram-matrix least squares with Tikhonov regularisation, AA point every E iterations,
verification one iteration later (no extra map evaluation on accept)."""
import sys, os, time, math
sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/scratch")
import numpy as np, omc_oracle as orc
import aa_experiment as ex

def run(inst, cuts, cut_type, rho_scale, mode, mem=5, every=5, start=50, reg=1e-10, max_iters=3000, safeguard=1.0):
    ex.inst = inst
    S = ex.Solver.__new__(ex.Solver)
    # generic cut type support
    p = orc.RelaxParams(rho_scale=rho_scale); S.inst = inst; S.p = p
    n, m, k, g = inst.n, inst.m, inst.k, inst.gamma
    S.rows = orc.build_rows(inst, cuts, cut_type, None, None, True); R = len(S.rows)
    S.Q = orc.row_subspace(S.rows, n, k); S.r = S.Q.shape[1]
    S.rho = orc.initial_rho(inst, p); S.rx = 1.6
    S.wY1 = p.rho_f_ratio * inst.N + 2.0
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); S.b = np.array(S.rows.rhs)
    for rr in range(R):
        if S.rows.kinds[rr] == "trace": AY[rr] = np.eye(n).ravel()
        elif S.rows.kinds[rr] == "cut": AY[rr] = np.outer(S.rows.xs[rr], S.rows.xs[rr]).ravel()
        AU[rr] = S.rows.CU[rr].ravel()
    S.AY, S.AU = AY, AU
    S.G1 = (AY / S.wY1.ravel()) @ AY.T + (AU / 2.0) @ AU.T
    S.svals = [None] * len(inst.groups)
    s = S.init_state(); best_lb = -1e300
    hist_f, hist_g = [], []
    zin = S.pack(s); pending = None; nacc = nrej = 0; evals = 0; gap = 1.0
    for it in range(1, max_iters + 1):
        g_ = S.step(s); evals += 1
        gz = S.pack(g_); f = gz - zin; fn = np.linalg.norm(f)
        if mode == "aa" and it >= start:
            if pending is not None:
                if fn <= safeguard * pending["fn"]:
                    nacc += 1
                else:                        # reject: back to the image of the last plain step, forget the history
                    nrej += 1
                    s = S.unpack(pending["g"], g_); zin = pending["g"].copy(); hist_f, hist_g = [], []; pending = None
                    continue
                pending = None
            hist_f.append(f); hist_g.append(gz)
            if len(hist_f) > mem + 1: hist_f.pop(0); hist_g.pop(0)
            if it % every == 0 and len(hist_f) >= 3:
                F = np.array(hist_f); G = np.array(hist_g)
                dF = (F[1:] - F[:-1]); dG = (G[1:] - G[:-1])
                H = dF @ dF.T; rhs = dF @ f
                H = H + reg * np.trace(H) / len(H) * np.eye(len(H))
                try:
                    gam = np.linalg.solve(H, rhs)
                except np.linalg.LinAlgError:
                    gam = None
                if gam is not None and np.isfinite(gam).all():
                    zaa = gz - dG.T @ gam
                    pending = dict(fn=fn, g=gz)
                    s = S.unpack(zaa, g_); zin = zaa
                    # symmetrise matrices (AA keeps symmetry automatically: linear combination of symmetric matrices)
                    continue_flag = True
                else:
                    s = g_; zin = gz
            else:
                s = g_; zin = gz
        else:
            s = g_; zin = gz
        if it % 25 == 0:
            obj, lb = S.certificate(s if pending is None else S.unpack(pending["g"], g_)) if False else S.certificate(g_)
            best_lb = max(best_lb, lb); gap = (obj - best_lb) / max(1.0, abs(obj))
            if gap <= 1e-6 and S.rp <= 1e-7 * math.sqrt(n + k):
                break
    return dict(iters=it, gap=gap, rp=S.rp, acc=nacc, rej=nrej)

if __name__ == "__main__":
    cases = []
    A, mask = orc.make_instance(20, 24, 1, seed=11, kind="readme"); i1 = orc.Instance(A, mask, 80.0, 1)
    slow = list(np.load("/root/repo/scratch/slow_nodes.npy", allow_pickle=True)[0]["cuts"])
    cases.append(("readme20x24 slow node", i1, slow, "linear", 16.0))
    cases.append(("readme20x24 root", i1, [], "linear", 16.0))
    # golden paths
    for f in ("readme_20x24_k1_linear", "lowrank_24x28_k1_linear2", "lowrank_16x20_k2_linear3"):
        z = np.load("/root/repo/tests/golden/%s.npz" % f, allow_pickle=False)
        DN = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
        cuts = [(z["cut_x"][l], z["cut_U"][l], [DN[int(c)] for c in z["cut_dir"][l]]) for l in range(len(z["cut_x"]))]
        inst = orc.Instance(z["A"], z["mask"], 80.0, int(z["k"]))
        for L in z["node_L"]:
            cases.append(("%s L=%d (golden iters %d)" % (f, L, z["iters"][list(z["node_L"]).index(L)]), inst, cuts[:int(L)], str(z["cut_type"]), float(z["rho_scale"])))
    for name, inst, cuts, ct, rs in cases:
        t0 = time.time()
        a = run(inst, cuts, ct, rs, "plain")
        b = run(inst, cuts, ct, rs, "aa", mem=5, every=5)
        c = run(inst, cuts, ct, rs, "aa", mem=10, every=5)
        d = run(inst, cuts, ct, rs, "aa", mem=10, every=10)
        print("%-52s plain %4d (gap %.0e) | aa5/5 %4d acc %d rej %d | aa10/5 %4d acc %d rej %d | aa10/10 %4d acc %d rej %d  (%.0fs)" % (
            name, a["iters"], a["gap"], b["iters"], b["acc"], b["rej"], c["iters"], c["acc"], c["rej"], d["iters"], d["acc"], d["rej"], time.time() - t0), flush=True)
