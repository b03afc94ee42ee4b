"""CPU tests: pin the oracle by closed-form known answers and by optimality certificates.
The reference ships no tests or fixtures (test/runtests.jl is empty) and cannot run here (Julia + Mosek), so these
KATs -- derived from the reference's own mathematics -- are what pins the oracle ("parity unpinned" vs Mosek)."""
import itertools

import numpy as np
import pytest
import scipy.sparse.linalg as spla

import omc_oracle as orc

GAMMA = 80.0


def fully_observed(n, m, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, m)), np.ones((n, m), bool)


def test_kat1_fully_observed_master_optimum():
    """indices all true: rank-k optimum X* = sum_{i<=k} s_i g/(1+g) u_i v_i', objective 1/2 sum_{i<=k} s_i^2/(1+g) + 1/2 sum_{i>k} s_i^2."""
    n, m, k = 12, 15, 2
    A, mask = fully_observed(n, m, 0)
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    Xs = (U[:, :k] * (s[:k] * GAMMA / (1 + GAMMA))) @ Vt[:k]
    expect = 0.5 * (s[:k] ** 2).sum() / (1 + GAMMA) + 0.5 * (s[k:] ** 2).sum()
    assert orc.evaluate_objective(Xs, A, mask, GAMMA) == pytest.approx(expect, rel=1e-13)
    # unconstrained altmin from the SVD start converges to it (box/ball inactive at the orthonormal start only up to scale,
    # so compare objective values, which are scale invariant in U V)
    inst = orc.Instance(A, mask, GAMMA, 1)
    am = orc.alternating_minimization(inst, U[:, :1])
    expect1 = 0.5 * s[0] ** 2 / (1 + GAMMA) + 0.5 * (s[1:] ** 2).sum()
    assert orc.evaluate_objective(am["U"] @ am["V"], A, mask, GAMMA) == pytest.approx(expect1, rel=1e-6)


@pytest.mark.parametrize("k", [1, 2])
def test_kat2_fully_observed_root_relaxation_is_waterfilling(k):
    """Root relaxation with full observation = min sum_i 1/2 s_i^2/(1+g y_i), 0<=y<=1, sum y<=k (1-D water filling)."""
    from scipy.optimize import minimize
    n, m = 10, 12
    A, mask = fully_observed(n, m, 1)
    s = np.linalg.svd(A, compute_uv=False)
    obj = lambda y: float((0.5 * s ** 2 / (1 + GAMMA * y)).sum())
    res = minimize(obj, np.full(n, k / n), bounds=[(0, 1)] * n, method="SLSQP",
                   constraints=[dict(type="ineq", fun=lambda y: k - y.sum())], options=dict(ftol=1e-15, maxiter=1000))
    inst = orc.Instance(A, mask, GAMMA, k)
    r = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=16.0, eps_gap=1e-7))
    assert r["termination_status"] == orc.OMC_OPTIMAL
    assert r["objective"] == pytest.approx(res.fun, rel=2e-7)
    assert r["dual_bound"] <= r["objective"] + 1e-9
    assert r["objective"] - r["dual_bound"] <= 1e-7 * abs(r["objective"]) * 1.01


def test_kat3_v_step_is_exact_minimiser():
    rng = np.random.default_rng(2)
    A, mask = orc.make_instance(14, 18, 2, n_indices=120, seed=2)
    inst = orc.Instance(A, mask, GAMMA, 2)
    U = rng.standard_normal((14, 2)) * 0.3
    V = orc.altmin_v_step(inst, U)
    f0 = orc.altmin_objective(inst, U, V)
    for _ in range(20):
        assert orc.altmin_objective(inst, U, V + 1e-4 * rng.standard_normal(V.shape)) >= f0 - 1e-13


def test_kat4_partial_minimisation_identity():
    """f(Y) = min_X 1/2 sum_Omega (A-X)^2 + tr(X' pinv(Y) X)/(2 gamma); X = gamma Y[:,O] alpha; finite-difference gradient."""
    rng = np.random.default_rng(3)
    n, m = 9, 11
    A, mask = orc.make_instance(n, m, 1, n_indices=50, seed=3)
    inst = orc.Instance(A, mask, GAMMA, 1)
    B = rng.standard_normal((n, n)); Y = B @ B.T; Y /= np.trace(Y)
    f, Lam = inst.f_value(Y, want=True)
    X = inst.X_of(Y, Lam)
    Theta = X.T @ np.linalg.pinv(Y) @ X
    assert orc.compute_SDP_relaxation_objective(X, Theta, A, mask, GAMMA) == pytest.approx(f, rel=1e-10)
    assert np.allclose(Theta, GAMMA * Lam.T @ X, atol=1e-9)
    # direct per-column minimisation
    Yi = np.linalg.inv(Y); tot = 0.0
    for j in range(m):
        o = np.flatnonzero(mask[:, j]); P = np.zeros((n, n)); P[o, o] = 1.0
        x = np.linalg.solve(P + Yi / GAMMA, P @ A[:, j])
        tot += 0.5 * ((A[o, j] - x[o]) ** 2).sum() + x @ Yi @ x / (2 * GAMMA)
        assert np.allclose(x, X[:, j], atol=1e-8)
    assert tot == pytest.approx(f, rel=1e-9)
    G = -0.5 * GAMMA * Lam @ Lam.T
    D = rng.standard_normal((n, n)); D = D + D.T
    h = 1e-6
    fd = (inst.f_value(Y + h * D) - inst.f_value(Y - h * D)) / (2 * h)
    assert fd == pytest.approx(float((G * D).sum()), rel=1e-6)


def test_kat5_eigen_oracle_matches_arpack():
    """Dense eigh vs ARPACK eigsh(which='SA', tol=1e-6) -- the algorithm behind the reference's eigs(...; which=:SR) (OMC.jl:1274, 2467)."""
    rng = np.random.default_rng(4)
    n = 40
    B = rng.standard_normal((n, n)); Y = B @ B.T / n; u = rng.standard_normal((n, 1)) * 0.2
    x, ev = orc.breakpoint_vector(Y, u)
    w, v = spla.eigsh(u @ u.T - Y, k=1, which="SA", tol=1e-6)
    assert ev[0] == pytest.approx(w[0], abs=1e-6)
    assert min(np.linalg.norm(x - v[:, 0]), np.linalg.norm(x + v[:, 0])) < 1e-4
    assert x[np.argmax(np.abs(x))] > 0          # sign convention
    x2, ev2 = orc.breakpoint_vector(Y, u, "smallest_2_eigvec")
    assert np.linalg.norm(x2) == pytest.approx(1.0, abs=1e-12)   # w1^2 + w2^2 = 1 (OMC.jl:2472)


def test_cut_piece_table_and_quirk_q1():
    for ct, dirs in orc.DIRECTIONS_OF.items():
        for vhat in (-0.7, -0.2, 0.0, 0.3, 0.9):
            pieces = [orc.cut_piece(ct, d, vhat) for d in dirs]
            # pieces tile [-1, 1]
            assert pieces[0][0] == -1.0 and pieces[-1][1] == 1.0
            for a, b in zip(pieces[:-1], pieces[1:]):
                assert a[1] == pytest.approx(b[0])
            for d, (lo, hi, sl, ic) in zip(dirs, pieces):
                if hi < lo:
                    continue
                vs = np.linspace(lo, hi, 7)
                g = sl * vs + ic
                if ct == "linear3" and d == "right":
                    assert np.all(g <= vs ** 2 + 1e-12)         # quirk Q1: a*v UNDER-estimates v^2 on [a,1] (OMC.jl:1675)
                    lo2, hi2, sl2, ic2 = orc.cut_piece(ct, d, vhat, reference_quirk_q1=False)
                    assert np.all(sl2 * vs + ic2 >= vs ** 2 - 1e-12)
                else:
                    assert np.all(g >= vs ** 2 - 1e-12)          # secant over-estimator of v^2 on the piece
    with pytest.raises(ValueError):
        orc.cut_piece("quadratic", "left", 0.1)


def test_child_directions_first_column_fastest():
    assert orc.child_directions("linear", 1) == [["left"], ["right"]]
    d = orc.child_directions("linear2", 2)
    assert len(d) == 9 and d[0] == ["left", "left"] and d[1] == ["middle", "left"] and d[3] == ["left", "middle"]
    assert len(orc.child_directions("linear3", 2)) == 16


def brute_shor(mask, p):
    n, m = mask.shape; out = set()
    for i1, i2 in itertools.combinations(range(n), 2):
        for j1, j2 in itertools.combinations(range(m), 2):
            cnt = int(mask[i1, j1]) + int(mask[i1, j2]) + int(mask[i2, j1]) + int(mask[i2, j2])
            if cnt == p:
                out.add((i1 + 1, i2 + 1, j1 + 1, j2 + 1))
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_shor_enumeration_against_brute_force(seed):
    rng = np.random.default_rng(seed)
    mask = rng.integers(0, 2, (6, 7)).astype(bool)
    for p in (0, 1, 3, 4):
        got = orc.shor_constraints_indexes(mask, [p])
        assert len(got) == len(set(got))
        assert set(got) == brute_shor(mask, p)
    # p = 2: the reference enumerates sub-case (a) [1 0;1 0]-type columns and (b) two xor columns (OMC.jl:2569-2583);
    # minors whose two observed cells are on a diagonal of two xor columns are covered by (b), same-row pairs are NOT
    got2 = set(orc.shor_constraints_indexes(mask, [2]))
    assert got2 <= brute_shor(mask, 2) | brute_shor(mask, 2)
    X3 = rng.standard_normal((1, 6, 7))
    top = orc.violated_shor_minors(X3, mask, [4], [], 5)
    assert len(top) <= 5 and all(top[i][0] >= top[i + 1][0] for i in range(len(top) - 1))


def path_nodes(inst, cut_type, depth, rho_scale, seed=5):
    rng = np.random.default_rng(seed)
    dirs = orc.child_directions(cut_type, inst.k)
    cuts = []; nodes = [[]]; results = []
    for d in range(depth):
        r = orc.sdp_relaxation(inst, cuts, cut_type, params=orc.RelaxParams(rho_scale=rho_scale))
        results.append(r)
        x, _ = orc.breakpoint_vector(r["Y"], r["U"])
        cuts = cuts + [(x, r["U"].copy(), dirs[int(rng.integers(len(dirs)))])]
        nodes.append(list(cuts))
    return nodes, results


@pytest.mark.parametrize("n,m,k,kind,cut_type,rho_scale", [
    (16, 18, 1, "readme", "linear", 16.0),
    (18, 20, 1, "lowrank", "linear2", 4.0),
    (14, 16, 2, "lowrank", "linear", 4.0),
])
def test_relaxation_certificates_and_nesting(n, m, k, kind, cut_type, rho_scale):
    """Every returned point is feasible for the reference's conic program (OMC.jl:1554-1685) up to tolerance, its
    objective equals the reference formula (OMC.jl:1970-1975), the dual bound certifies it, and child >= parent."""
    A, mask = orc.make_instance(n, m, k, seed=7, kind=kind, n_indices=None if kind == "readme" else int(0.4 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    nodes, results = path_nodes(inst, cut_type, 4, rho_scale)
    prev = -np.inf
    for r in results:
        if r["termination_status"] != orc.OMC_OPTIMAL:
            continue
        scale = max(1.0, np.abs(r["Theta"]).max())
        assert r["residuals"]["max"] <= 2e-5 * scale, r["residuals"]
        assert r["objective_reference_formula"] == pytest.approx(r["objective"], rel=1e-9)
        assert r["objective"] - r["dual_bound"] <= 1.01e-6 * max(1.0, abs(r["objective"]))
        assert r["dual_bound"] >= prev - 2e-6 * abs(r["objective"])      # feasible sets nest (OMC.jl:2522)
        prev = r["dual_bound"]


def test_infeasible_node_is_reported():
    """Contradictory cuts (v <= -0.5 and v >= 0.5 on the same direction) -> OMC_INFEASIBLE via an unbounded dual bound."""
    A, mask = orc.make_instance(12, 14, 1, seed=1, kind="readme")
    inst = orc.Instance(A, mask, GAMMA, 1)
    x = np.zeros(12); x[0] = 1.0
    Uh = np.zeros((12, 1)); Uh[0, 0] = 0.5
    cuts = [(x, Uh, ["right"]), (x, -Uh, ["left"])]
    r = orc.sdp_relaxation(inst, cuts, "linear", params=orc.RelaxParams(rho_scale=16.0, max_iters=3000), want_certificate=False)
    assert r["termination_status"] == orc.OMC_INFEASIBLE and not r["feasible"]


def test_altmin_u_step_kkt_and_quirk_q3():
    A, mask = orc.make_instance(15, 18, 1, n_indices=120, seed=9)
    inst = orc.Instance(A, mask, GAMMA, 1)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 1)
    x = np.linalg.qr(np.random.default_rng(0).standard_normal((15, 1)))[0][:, 0]
    cuts = [(x, U0 * 0.5, ["left"])]
    V = orc.altmin_v_step(inst, U0)
    Un, obj, info = orc.altmin_u_step(inst, V, cuts, "linear")
    u = Un[:, 0]
    lo, hi, _, _ = orc.cut_piece("linear", "left", float((U0 * 0.5).T @ x))
    assert u @ u <= 1 + 1e-9 and lo - 1e-9 <= x @ u <= hi + 1e-9 and u[-1] >= -1e-12
    # KKT: gradient + theta*u + C'lam = 0 with the returned multipliers
    H, gv, _ = orc._ustep_quadratic(inst, V)
    grad = H[:, 0, 0] * u - gv[:, 0] + info["theta"] * u
    rows = orc.build_rows(inst, cuts, "linear")
    sel = [r for r in range(len(rows)) if rows.kinds[r] in ("box_lo", "box_hi", "hi", "lo")]
    C = np.array([rows.CU[r].ravel() for r in sel])
    assert np.allclose(grad + C.T @ info["lam"], 0, atol=1e-8)
    assert obj == pytest.approx(orc.altmin_objective(inst, Un, V), rel=1e-10)
    am = orc.alternating_minimization(inst, U0, cuts, "linear")
    assert am["n_iters"] <= 100 and len(am["objectives"]) == am["n_iters"]
    if am["converged"] and am["n_iters"] > 1:
        o = am["objectives"]
        assert abs((o[-1] - o[-2]) / o[-2]) < 1e-5 or all(o[-1 - i] > o[-6] for i in range(5))   # OMC.jl:2234-2245


def test_objective_functions_and_errors():
    A, mask = orc.make_instance(8, 10, 1, seed=0, kind="readme")
    X = np.random.default_rng(1).standard_normal((8, 10))
    ref = 0.5 * sum((X[i, j] - A[i, j]) ** 2 for i in range(8) for j in range(10) if mask[i, j]) + (X ** 2).sum() / (2 * GAMMA)
    assert orc.evaluate_objective(X, A, mask, GAMMA) == pytest.approx(ref, rel=1e-13)
    assert orc.compute_MSE(X, A, mask, "all") == pytest.approx(((X - A) ** 2).mean())
    with pytest.raises(ValueError):
        orc.evaluate_objective(X[:, :5], A, mask, GAMMA)
    with pytest.raises(ValueError):
        orc.Instance(np.zeros((5, 3)), np.ones((5, 3), bool), GAMMA, 1)    # n <= m required (OMC.jl:249-254)
    with pytest.raises(ValueError):
        orc.compute_MSE(X, A, mask, "median")


def test_oracle_anderson_acceleration_converges_crawling_node(orc):
    """The README fixture's node L = 2 crawls (SLOW_PROGRESS); with accel = 1 the oracle certifies it in a few
    hundred iterations and returns the same optimum value."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_20x24_k1_linear.npz"), allow_pickle=False)
    DN = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
    cuts = [(z["cut_x"][l], z["cut_U"][l], [DN[int(c)] for c in z["cut_dir"][l]]) for l in range(2)]
    inst = orc.Instance(z["A"], z["mask"], 80.0, 1)
    r = orc.sdp_relaxation(inst, cuts, "linear", params=orc.RelaxParams(rho_scale=16.0, accel=1), want_certificate=False)
    assert int(z["status"][2]) == 1 and int(z["iters"][2]) >= 1000          # without acceleration (golden): SLOW_PROGRESS, stopped early once hopeless
    assert r["termination_status"] == 0 and r["iters"] <= 500 and r["n_aa"] > 10
    assert r["objective"] == pytest.approx(float(z["objective"][2]), rel=2e-6)
    assert r["objective"] - r["dual_bound"] <= 1e-6 * max(1.0, abs(r["objective"]))


def test_ustep_dual_newton_matches_slsqp(orc):
    """Rank k > 1 U-step of altmin: the exact dual Newton method (what the GPU mirrors) against scipy SLSQP on the same QP."""
    rng = np.random.default_rng(0)
    for trial in range(6):
        k = 2 if trial % 3 else 3
        n = int(rng.integers(8, 14)); m = n + int(rng.integers(0, 6))
        A, mask = orc.make_instance(n, m, k, seed=100 + trial, kind="lowrank", n_indices=max(int(0.6 * n * m), (n + m) * k))
        sc = float(10 ** rng.uniform(-1, 1))
        inst = orc.Instance(A * sc, mask, 80.0, k)
        V = rng.standard_normal((k, m)) * sc
        cuts = []
        for _ in range(int(rng.integers(0, 3))):
            x = rng.standard_normal(n); x /= np.linalg.norm(x)
            cuts.append((x, rng.standard_normal((n, k)) * 0.3, [["left", "right"][int(rng.integers(2))] for _ in range(k)]))
        U1, o1, i1 = orc.altmin_u_step(inst, V, cuts, "linear", method="dual")
        U2, o2, i2 = orc.altmin_u_step(inst, V, cuts, "linear", method="slsqp")
        W, rad = orc.quadratic_constraint_vectors(k)
        assert i1["kkt_residual"] <= 1e-9 and (((U1 @ W.T) ** 2).sum(0) - rad).max() <= 1e-9
        if i2["slsqp_status"] == 0:
            assert o1 == pytest.approx(o2, rel=1e-7) and o1 <= o2 + 1e-9 * max(1.0, abs(o2))


@pytest.mark.parametrize("seed,cut_type,dirs", [(0, "linear", ["right"]), (1, "linear", ["left"]), (2, "linear2", ["right"]), (3, "linear3", ["left"]), (4, "linear2", ["middle"])])
def test_dual_bound_against_an_independent_solve_on_a_cut_node(seed, cut_type, dirs):
    """Validity of the certificate on a node WITH a cut, checked by a different method: the node program restricted to its (Y, U) form
    (X, Theta eliminated in closed form, KAT-4) is solved by scipy SLSQP over the factorisation Y = U U' + R R' (so [Y U; U' I] >= 0 holds
    by construction), with I - Y >= 0 as an eigenvalue constraint and the cut rows of OMC.jl:1580-1683 written out by hand.  Every point
    SLSQP visits is feasible for the relaxation, so its value can never fall below the oracle's dual bound; at convergence it matches
    the oracle's objective."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(seed)
    n, m, k = 5, 6, 1
    A, mask = orc.make_instance(n, m, k, seed=40 + seed, kind="lowrank", n_indices=18, noise=0.3)
    inst = orc.Instance(A, mask, GAMMA, k)
    root = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=8.0), want_certificate=False)
    x, _ = orc.breakpoint_vector(root["Y"], root["U"])
    cut = (x, root["U"].copy(), dirs)
    r = orc.sdp_relaxation(inst, [cut], cut_type, params=orc.RelaxParams(rho_scale=8.0))
    assert r["termination_status"] in (orc.OMC_OPTIMAL, orc.OMC_SLOW_PROGRESS)      # the `middle` piece has no Slater point when |v-hat| is small: bound valid, not certified
    vhat = float(root["U"][:, 0] @ x)
    lo, hi, sl, ic = orc.cut_piece(cut_type, dirs[0], vhat)
    iu = np.tril_indices(n)

    def unpack(z):
        u = z[:n]; R = np.zeros((n, n)); R[iu] = z[n:]
        return u, np.outer(u, u) + R @ R.T

    def fobj(z):
        _, Y = unpack(z)
        return inst.f_value(Y)

    cons = [
        dict(type="ineq", fun=lambda z: 1.0 - np.trace(unpack(z)[1])),                                  # tr Y <= k      (OMC.jl:1558)
        dict(type="ineq", fun=lambda z: 1.0 - np.linalg.eigvalsh(unpack(z)[1])[-1]),                   # I - Y >= 0     (OMC.jl:1556)
        dict(type="ineq", fun=lambda z: 1.0 - float(z[:n] @ z[:n])),                                   # ||U_1|| <= 1   (OMC.jl:1831-1835)
        dict(type="ineq", fun=lambda z: z[n - 1]),                                                    # U[n, 1] >= 0   (OMC.jl:1442-1449)
        dict(type="ineq", fun=lambda z: float(x @ z[:n]) - lo), dict(type="ineq", fun=lambda z: hi - float(x @ z[:n])),   # bounds on v
        dict(type="ineq", fun=lambda z: sl * float(x @ z[:n]) + ic - float(x @ unpack(z)[1] @ x)),    # x'Yx <= g(v)   (OMC.jl:1680-1683)
    ]
    best = np.inf
    for trial in range(6):
        u0 = r["U"][:, 0] + 0.05 * rng.standard_normal(n) if trial == 0 else 0.3 * rng.standard_normal(n)
        z0 = np.concatenate([u0, 0.2 * rng.standard_normal(len(iu[0]))])
        res = minimize(fobj, z0, constraints=cons, bounds=[(-1, 1)] * n + [(None, None)] * len(iu[0]), method="SLSQP", options=dict(ftol=1e-13, maxiter=500))
        feas = all(c["fun"](res.x) >= -1e-8 for c in cons)
        if feas:
            assert r["dual_bound"] <= res.fun + 1e-7 * abs(res.fun)          # a feasible value can never undercut a valid bound
            best = min(best, res.fun)
    assert np.isfinite(best)
    if r["termination_status"] == orc.OMC_OPTIMAL:
        assert r["objective"] == pytest.approx(best, rel=2e-5)                # and the independent optimum is the oracle's
