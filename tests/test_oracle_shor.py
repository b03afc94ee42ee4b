"""CPU tests of the Shor-mode oracle (oracle/omc_oracle_shor.py): known answers derived from the reference's program
(OMC.jl:1503-1525, 1755-1779, 1838-1846) and the two-sided certificate.  Parity unpinned (no reference output exists): these are
what pins the restatement."""
import numpy as np
import pytest

import omc_oracle as orc
import omc_oracle_shor as sh

GAMMA = 80.0


def _inst(n, m, nidx, seed, noise, kind="lowrank"):
    A, mask = orc.make_instance(n, m, 1, n_indices=nidx, seed=seed, noise=noise, kind=kind)
    return orc.Instance(A, mask, GAMMA, 1)


def test_paraboloid_projection_is_the_nearest_point():
    rng = np.random.default_rng(0)
    for _ in range(20):
        xi = rng.standard_normal(5); t = float(rng.standard_normal())
        px, pt, nu = sh.proj_paraboloid(xi, t)
        assert pt >= px @ px - 1e-12
        if nu > 0:
            assert pt == pytest.approx(px @ px, abs=1e-10)
            # stationarity: (xi - px, t - pt) is a negative multiple of the outward normal (2 px, -1)
            assert np.allclose(xi - px, 2.0 * nu * px, atol=1e-10) and t - pt == pytest.approx(-nu, abs=1e-12)
        # no feasible point is closer
        d0 = (xi - px) @ (xi - px) + (t - pt) ** 2
        for _ in range(30):
            y = rng.standard_normal(5); s = float(y @ y + abs(rng.standard_normal()))
            assert (xi - y) @ (xi - y) + (t - s) ** 2 >= d0 - 1e-12


def test_no_minors_equals_the_base_relaxation():
    """With an empty minor list and the SOC list on every entry (the reference's iterative-mode root, OMC.jl:670-674) W = X^2 on the
    observed entries and the slack of Theta_jj = sum_i W_ij goes to an unobserved entry: the value is the base relaxation's."""
    inst = _inst(10, 12, 60, 1, 0.3)
    assert (~inst.indices).any(0).all()                     # every column has an unobserved entry
    minors, soc = sh.driver_shor_lists(inst.indices, minors=[])
    r = sh.sdp_relaxation_shor(inst, minors, soc)
    b = orc.sdp_relaxation(inst)
    assert r["termination_status"] == orc.OMC_OPTIMAL and b["termination_status"] == orc.OMC_OPTIMAL
    assert r["objective"] == pytest.approx(b["objective"], rel=3e-6)
    assert r["dual_bound"] <= r["objective"] + 1e-6 * abs(r["objective"])


def test_fully_observed_column_costs_more_than_the_base_relaxation():
    """A fully observed column has no unobserved entry to carry the slack of Theta_jj = sum_i W_ij, so there W sits above X^2 on
    observed entries (cost 1/2 per unit, OMC.jl:1842): the Shor value is >= the base value, and the certificate still closes."""
    A, mask = orc.make_instance(8, 10, 1, n_indices=50, seed=5, noise=0.3)
    mask[:, 2] = True
    inst = orc.Instance(A, mask, GAMMA, 1)
    minors, soc = sh.driver_shor_lists(inst.indices, minors=[])
    r = sh.sdp_relaxation_shor(inst, minors, soc)
    b = orc.sdp_relaxation(inst)
    assert r["structure"].ctype[2] == 1
    assert r["termination_status"] == orc.OMC_OPTIMAL
    assert r["objective"] >= b["objective"] - 1e-6
    assert r["residuals"]["max"] <= 1e-6
    assert r["objective_reference_formula"] == pytest.approx(r["objective"], rel=1e-9)


@pytest.mark.parametrize("seed,noise,eps", [(2, 0.1, 1e-6), (4, 0.2, 1e-5)])
def test_static_class4_minors_sandwich_and_certificate(seed, noise, eps):
    """base relaxation <= Shor relaxation <= master optimum (any rank-1 X with W = X^2, Theta = X'X, Y = uu' is feasible: the value
    of evaluate_objective at the altmin point bounds it from above); primal residuals of every cone of the reference's program."""
    inst = _inst(12, 14, 70, seed, noise)
    minors, soc = sh.driver_shor_lists(inst.indices, (4,))
    assert len(minors) > 50
    r = sh.sdp_relaxation_shor(inst, minors, soc, params=sh.ShorParams(eps_gap=eps))
    b = orc.sdp_relaxation(inst)
    U0 = orc.svd_rounding(np.where(inst.indices, inst.A, 0.0), 1)
    am = orc.alternating_minimization(inst, U0)
    ub = orc.evaluate_objective(am["U"] @ am["V"], inst.A, inst.indices, GAMMA)
    assert r["termination_status"] == orc.OMC_OPTIMAL
    assert b["objective"] - 1e-6 <= r["objective"] <= ub + 1e-6
    assert r["objective"] > b["objective"] + 1e-3            # the minors cut something off on noisy data
    assert abs(r["objective"] - r["dual_bound"]) <= eps * max(1.0, abs(r["objective"])) * 1.01
    assert r["residuals"]["max"] <= 1e-6
    assert r["objective_reference_formula"] == pytest.approx(r["objective"], rel=1e-9)   # OMC.jl:1960-1967 on the returned (X, W, Theta)


def test_rank_one_data_sandwich_with_the_planted_matrix():
    """Noise-free rank-1 data A = l r': every multiple c A has vanishing 2 x 2 minors, so (X, W, Theta, Y) = (cA, X^2, X'X, uu') is feasible
    for the Shor program and its value evaluate_objective(cA) bounds the relaxation from above for every c; the base relaxation bounds
    it from below.  (The relaxed X itself may have rank 2: only the fully observed minors are constrained.)"""
    inst = _inst(10, 12, 70, 3, 0.0)
    minors, soc = sh.driver_shor_lists(inst.indices, (4,))
    r = sh.sdp_relaxation_shor(inst, minors, soc, params=sh.ShorParams(eps_gap=1e-6))
    b = orc.sdp_relaxation(inst)
    assert r["termination_status"] == orc.OMC_OPTIMAL
    ub = min(orc.evaluate_objective(c * inst.A, inst.A, inst.indices, GAMMA) for c in np.linspace(0.5, 1.0, 501))
    assert b["objective"] - 1e-6 <= r["objective"] <= ub + 1e-6
    Xs = r["X"]
    worst = max(abs(Xs[i1 - 1, j1 - 1] * Xs[i2 - 1, j2 - 1] - Xs[i1 - 1, j2 - 1] * Xs[i2 - 1, j1 - 1]) for (i1, i2, j1, j2) in minors)
    assert worst <= 1e-4 * np.abs(Xs).max() ** 2            # the minors it does constrain vanish on this data


def test_minors_with_cuts_bound_is_valid_and_above_the_parent():
    """A depth-2 cut node with the static minors: the dual bound is a valid lower bound (below every feasible value: here the
    node's own primal value) and the value is >= the same node without minors and >= the root with minors (feasible sets nest)."""
    inst = _inst(12, 14, 70, 2, 0.1)
    minors, soc = sh.driver_shor_lists(inst.indices, (4,))
    cuts = []
    for d in range(2):
        r0 = orc.sdp_relaxation(inst, cuts=cuts)
        x, _ = orc.breakpoint_vector(r0["Y"], r0["U"])
        cuts = cuts + [(x, r0["U"], ["left" if d % 2 == 0 else "right"])]
    root = sh.sdp_relaxation_shor(inst, minors, soc)
    r = sh.sdp_relaxation_shor(inst, minors, soc, cuts=cuts, params=sh.ShorParams(max_iters=3000))
    b = orc.sdp_relaxation(inst, cuts=cuts)
    assert r["dual_bound"] <= r["objective"] + 1e-5
    assert r["objective"] - r["dual_bound"] <= 1e-3 * r["objective"]
    assert r["objective"] >= b["objective"] - 1e-6
    assert r["objective"] >= root["objective"] - 1e-5
    assert r["residuals"]["max"] <= 1e-4


def test_dual_bound_is_valid_for_arbitrary_multipliers():
    """The bound must hold for ANY multipliers in their cones -- feed random ones and compare with a converged value."""
    inst = _inst(8, 9, 40, 4, 0.2)
    minors, soc = sh.driver_shor_lists(inst.indices, (4,))
    r = sh.sdp_relaxation_shor(inst, minors, soc)
    assert r["termination_status"] == orc.OMC_OPTIMAL
    st = r["structure"]; sc = r["scale"]; n, m = inst.n, inst.m
    rng = np.random.default_rng(0)
    mask = inst.indices
    qX = np.where(mask & st.inS & (st.ctype == 0)[None, :], 1.0, 0.0)
    cX = np.where(mask, -inst.A * sc, 0.0)
    cW = np.where(mask & st.inC, 0.5, 0.0) - np.where(st.inC & (st.ctype == 1)[None, :], 0.5, 0.0)
    cT = 1.0 / (2.0 * GAMMA) + np.where(st.ctype == 1, 0.5, 0.0)
    rows = r["rows"]; Q = np.zeros((n, 0))
    for trial in range(20):
        B = rng.standard_normal((st.nq, 5, 5)) * 0.02 * rng.random()
        Gam = np.einsum("qij,qkj->qik", B, B)
        zeta = rng.random(m) * 0.003
        xi = rng.standard_normal((n, m)) * 0.3
        Xbar = rng.standard_normal((n, m)) * 0.3
        lam = np.zeros(len(rows))
        lbv = sh.shor_dual_bound(inst, st, inst.A * sc, rows, lam, Q, np.zeros((1, 1)), Gam, zeta, xi, Xbar, qX, cX, cW, cT) / sc ** 2
        assert lbv <= r["objective"] + 1e-6 * abs(r["objective"])


def test_structure_rejects_malformed_lists():
    mask = np.ones((4, 5), bool)
    with pytest.raises(ValueError):
        sh.ShorStructure(4, 5, [(2, 1, 1, 2)], [], mask)          # i1 < i2 required
    with pytest.raises(ValueError):
        sh.ShorStructure(4, 5, [(1, 2, 1, 6)], [], mask)          # column out of range
    with pytest.raises(ValueError):
        sh.ShorStructure(4, 5, [(1, 2, 1, 2), (1, 2, 1, 2)], [], mask)
    with pytest.raises(ValueError):
        sh.ShorStructure(4, 5, [], [(5, 1)], mask)


def test_slow_node_returns_values_and_a_valid_bound():
    """A noisy instance on which the splitting crawls: the node comes back SLOW_PROGRESS (MOI.SLOW_PROGRESS with values, OMC.jl:1871-1877)
    with a bound that is still below every feasible value and above the base relaxation's certified bound minus tolerance."""
    inst = _inst(12, 14, 70, 7, 0.3)
    minors, soc = sh.driver_shor_lists(inst.indices, (4,))
    r = sh.sdp_relaxation_shor(inst, minors, soc, params=sh.ShorParams(max_iters=1500))
    U0 = orc.svd_rounding(np.where(inst.indices, inst.A, 0.0), 1)
    am = orc.alternating_minimization(inst, U0)
    ub = orc.evaluate_objective(am["U"] @ am["V"], inst.A, inst.indices, GAMMA)
    assert r["termination_status"] in (orc.OMC_SLOW_PROGRESS, orc.OMC_OPTIMAL) and r["feasible"]
    assert r["dual_bound"] <= ub + 1e-6
    assert r["objective"] - r["dual_bound"] <= 5e-3 * r["objective"]


def test_rank2_shor_minors_are_vacuous_and_the_extension_is_feasible():
    """Reference quirk Q5 (k > 1 form, OMC.jl:1526-1551, 1780-1827): the value with all class-4 minors equals the value without any (and the
    base relaxation's, every column having an unobserved entry), and the explicit extension (Xt, Wt, H, V) of the solution satisfies the
    reference's full constraint set: X = sum Xt, W = sum Wt + 2 sum H, every per-layer order-5 block, every per-coordinate order-3 block."""
    A, mask = orc.make_instance(10, 12, 2, n_indices=70, seed=6, noise=0.2)
    inst = orc.Instance(A, mask, GAMMA, 2)
    minors, soc = sh.driver_shor_lists(mask, (4,))
    assert len(minors) > 100
    r = sh.sdp_relaxation_shor(inst, minors, soc, params=sh.ShorParams(eps_gap=1e-6))
    r0 = sh.sdp_relaxation_shor(inst, [], sh.driver_shor_lists(mask, minors=[])[1], params=sh.ShorParams(eps_gap=1e-6))
    b = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=4.0))
    assert r["termination_status"] == orc.OMC_OPTIMAL and r0["termination_status"] == orc.OMC_OPTIMAL and b["termination_status"] == orc.OMC_OPTIMAL
    assert r["objective"] == pytest.approx(r0["objective"], rel=1e-9)        # literally the same program after the reduction
    assert r["objective"] == pytest.approx(b["objective"], rel=3e-6)
    assert r["residuals_rank_k"]["max"] <= 1e-10, r["residuals_rank_k"]
    # the extension really needs its slack: with s = 0 some layer-1 block of a minor with a non-vanishing determinant is indefinite
    ext = sh.complete_shor_rank_k(2, r["structure_full"], r["X"], r["W"])
    ext0 = dict(ext); ext0["Wt"] = ext["Wt"].copy(); ext0["Wt"][0] = np.where(r["structure_full"].inC, r["X"] ** 2, 0.0); ext0["Wt"][1] = 0.0; ext0["H"] = np.zeros_like(ext["H"])
    assert sh.shor_rank_k_residuals(2, r["structure_full"], r["X"], r["W"], ext0)["layer_minors"] > 1e-6
