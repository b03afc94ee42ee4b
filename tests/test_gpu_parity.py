"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Everything goes through the C ABI (ctypes), the
checker is the CPU oracle on the same seeded inputs.  Tolerances: the relaxation objective is compared at 2e-6
relative (both sides certify their value within 1e-6 of the optimum of the convex program); plain kernels
(objective scan, eigen-oracle) at 1e-12 / 1e-8."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GAMMA = 80.0
OBJ_REL = 2e-6


@pytest.fixture(scope="module")
def have_gpu(omc):
    lib = omc.load()
    if lib.omc_device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box (the HIP path has no CPU fallback)")
    return True


def oracle_path(orc, inst, cut_type, depth, rho_scale, seed, breakpoints="smallest_1_eigvec", q1=True, avoid=()):
    rng = np.random.default_rng(seed)
    dirs = orc.child_directions(cut_type, inst.k)
    cuts = []; nodes = [[]]
    for d in range(depth):
        r = orc.sdp_relaxation(inst, cuts, cut_type, params=orc.RelaxParams(rho_scale=rho_scale, reference_quirk_q1=q1), want_certificate=False)
        x, _ = orc.breakpoint_vector(r["Y"], r["U"], breakpoints)
        vhat = r["U"].T @ x
        ok = [d_ for d_ in dirs if all(((abs(vhat[j]) > 0.05) or (d_[j] in ("left", "right"))) and d_[j] not in avoid for j in range(inst.k))]
        cuts = cuts + [(x, r["U"].copy(), ok[int(rng.integers(len(ok)))])]
        nodes.append(list(cuts))
    return nodes


def assert_finite(o):
    """No NaN / Inf in anything a relaxation returns (an uninitialised pad once produced 0 x garbage = NaN in one run out of four)."""
    for key in ("objective", "dual_bound"):
        assert np.isfinite(o[key]) or (o["status_code"] == 3 and key == "objective"), (key, o[key])
    for key in ("U", "Y", "X", "Theta", "breakpoint_vec", "lambda_min"):
        if key in o and o["status_code"] != 3:
            assert np.isfinite(o[key]).all(), key


@pytest.mark.parametrize("n,m,k,kind,cut_type,rho_scale,depth", [
    (12, 15, 1, "readme", "linear", 8.0, 2),
    (20, 25, 1, "readme", "linear", 16.0, 4),
    (24, 30, 1, "lowrank", "linear2", 4.0, 3),
    (24, 24, 1, "lowrank", "linear3", 4.0, 3),
    (16, 20, 2, "lowrank", "linear", 4.0, 2),
    (50, 50, 1, "readme", "linear", 8.0, 2),       # BASELINE config 1 shape
])
def test_relaxation_matches_oracle(have_gpu, omc, orc, n, m, k, kind, cut_type, rho_scale, depth):
    A, mask = orc.make_instance(n, m, k, seed=21, kind=kind, n_indices=None if kind == "readme" else int(0.35 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    nodes = oracle_path(orc, inst, cut_type, depth, rho_scale, seed=3)
    ref = [orc.sdp_relaxation(inst, c, cut_type, params=orc.RelaxParams(rho_scale=rho_scale)) for c in nodes]
    eng = omc.Engine(A, mask, GAMMA, k)
    out = eng.matrix_completion_SDP_relaxation(nodes, cut_type, params=omc.default_params(rho_scale=rho_scale), want_Theta=True)
    for g, r in zip(out, ref):
        assert g["status_code"] == r["termination_status"]
        assert_finite(g)
        if g["status_code"] == 3:                      # infeasible node: no primal values (OMC.jl:1921-1935)
            assert not g["feasible"] and g["dual_bound"] > 0.5 * inst.sumA2
            continue
        assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
        if g["status_code"] == 0:
            assert abs(g["objective"] - g["dual_bound"]) <= 1.01e-6 * max(1.0, abs(g["objective"]))   # two-sided certificate
            assert g["dual_bound"] == pytest.approx(r["dual_bound"], rel=OBJ_REL)
        # separation oracle (OMC.jl:2466-2477): eigenvalue and canonical-sign eigenvector of U U' - Y
        S = g["U"] @ g["U"].T - g["Y"]
        w, V = np.linalg.eigh(0.5 * (S + S.T))
        assert g["lambda_min"][0] == pytest.approx(w[0], abs=1e-9)
        if w[1] - w[0] > 1e-6:
            v = V[:, 0] * np.sign(V[np.argmax(np.abs(V[:, 0])), 0])
            assert np.allclose(g["breakpoint_vec"], v, atol=1e-6)
        if g["status_code"] != 0:
            continue                                   # SLOW_PROGRESS: values available but not certified
        # the returned point is feasible for the reference's program (OMC.jl:1554-1685) and reproduces its objective
        Theta = 0.5 * (g["Theta"] + g["Theta"].T)
        res = orc.primal_residuals(inst, r["rows"], g["Y"], g["U"], g["X"], Theta)
        assert res["max"] <= 2e-5 * max(1.0, np.abs(Theta).max()), res
        assert orc.compute_SDP_relaxation_objective(g["X"], Theta, A, mask, GAMMA) == pytest.approx(g["objective"], rel=1e-8)
    eng.close()


def test_objective_scan_and_bitmatrix_constructor(have_gpu, omc, orc):
    rng = np.random.default_rng(0)
    n, m = 37, 53
    A, mask = orc.make_instance(n, m, 1, seed=1, kind="readme")
    eng = omc.Engine(A, mask, GAMMA, 1)
    Xs = rng.standard_normal((5, n, m))
    got = eng.evaluate_objective(Xs)
    ref = np.array([orc.evaluate_objective(x, A, mask, GAMMA) for x in Xs])
    assert np.allclose(got, ref, rtol=1e-12)
    with pytest.raises(ValueError):
        eng.evaluate_objective(np.zeros((n, m + 1)))                      # OMC.jl:2337-2348
    eng.close()
    # Julia BitMatrix layout: packed UInt64 chunks, column-major, LSB first
    bits = np.asfortranarray(mask).ravel(order="F")
    chunks = np.zeros((bits.size + 63) // 64, dtype=np.uint64)
    for e in np.flatnonzero(bits):
        chunks[e >> 6] |= np.uint64(1) << np.uint64(e & 63)
    lib = omc.load(); h = C.c_void_p(); Af = np.asfortranarray(A)
    rc = lib.omc_instance_create_bits(n, m, 1, Af.ctypes.data_as(C.c_void_p), chunks.ctypes.data_as(C.c_void_p), GAMMA, 0, C.byref(h))
    assert rc == 0
    out = np.zeros(1); X0 = np.ascontiguousarray(np.asfortranarray(Xs[0]).ravel(order="F"))
    assert lib.omc_evaluate_objective(h, 1, X0.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    assert out[0] == pytest.approx(ref[0], rel=1e-12)
    lib.omc_instance_destroy(h)


def test_separation_and_master_feasibility(have_gpu, omc, orc):
    rng = np.random.default_rng(5)
    n, m, k = 30, 32, 2
    A, mask = orc.make_instance(n, m, k, seed=2, kind="lowrank", n_indices=400)
    eng = omc.Engine(A, mask, GAMMA, k)
    Ys, Us = [], []
    for t in range(4):
        B = rng.standard_normal((n, n)); Y = B @ B.T / n; U = rng.standard_normal((n, k)) * 0.2
        Ys.append(Y); Us.append(U)
    Q = np.linalg.qr(rng.standard_normal((n, k)))[0]
    Ys.append(Q @ Q.T); Us.append(Q)                                     # master feasible: Y = U U'
    for bp in ("smallest_1_eigvec", "smallest_2_eigvec"):
        x, ev, fe = eng.breakpoint_vectors(Ys, Us, bp)
        for b in range(5):
            xr, evr = orc.breakpoint_vector(Ys[b], Us[b], bp)
            assert np.allclose(ev[b], evr, atol=1e-10)
            if b < 4:
                assert np.allclose(x[b], xr, atol=1e-7)
            assert bool(fe[b]) == orc.master_feasible(Ys[b], Us[b])[0]
    assert eng.matrix_completion_master_feasible(Ys[4], Us[4]) and not eng.matrix_completion_master_feasible(Ys[0], Us[0])
    with pytest.raises(ValueError):
        eng.breakpoint_vectors(Ys, Us, "largest_eigvec")                 # OMC.jl:2440-2446
    eng.close()


def test_error_behaviour_and_edge_cases(have_gpu, omc, orc):
    A, mask = orc.make_instance(10, 12, 1, seed=3, kind="readme")
    eng = omc.Engine(A, mask, GAMMA, 1)
    with pytest.raises(ValueError):
        eng.matrix_completion_SDP_relaxation([[]], "quadratic")           # OMC.jl:1456-1462
    x = np.zeros(10); x[0] = 1.0
    with pytest.raises(omc.OmcError) as e:
        eng.matrix_completion_SDP_relaxation([[(x, np.zeros((10, 1)), ["middle"])]], "linear")   # direction invalid for the type
    assert e.value.code == -1
    # contradictory cuts -> INFEASIBLE (OMC.jl:1921-1935: feasible = false)
    Uh = np.zeros((10, 1)); Uh[0, 0] = 0.5
    out = eng.matrix_completion_SDP_relaxation([[(x, Uh, ["right"]), (x, -Uh, ["left"])]], "linear", params=omc.default_params(rho_scale=16.0))
    assert out[0]["termination_status"] == "INFEASIBLE" and not out[0]["feasible"]
    # a column / row pattern with an empty column is legal input for the relaxation (no observed entries in column 0)
    mask2 = mask.copy(); mask2[:, 0] = False
    eng2 = omc.Engine(A, mask2, GAMMA, 1)
    inst2 = orc.Instance(A, mask2, GAMMA, 1)
    g = eng2.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=8.0))[0]
    r = orc.sdp_relaxation(inst2, params=orc.RelaxParams(rho_scale=8.0), want_certificate=False)
    assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    eng.close(); eng2.close()


def test_full_size_config2_properties(have_gpu, omc):
    """BASELINE config 2 shape (100x100, k=1, 20% observed): the oracle would take minutes, so check the
    size-independent properties: certified gap, bound <= objective, children >= parent, primal feasibility of Y."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, c["k"])
    P = omc.default_params(rho_scale=4.0)
    root = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P)[0]
    assert root["status_code"] == 0
    assert 0 <= root["objective"] - root["dual_bound"] + 1e-6 * root["objective"] <= 2.01e-6 * root["objective"]
    w = np.linalg.eigvalsh(root["Y"])
    assert w[0] >= -1e-6 and w[-1] <= 1 + 1e-6 and np.trace(root["Y"]) <= 1 + 1e-9
    S = np.block([[root["Y"], root["U"]], [root["U"].T, np.eye(1)]])
    assert np.linalg.eigvalsh(S)[0] >= -1e-6
    kids = omc.pkg.bnb.make_children([], root, "linear", 1)
    out = eng.matrix_completion_SDP_relaxation(kids, "linear", params=P, want_X=False)
    for o in out:
        if o["status_code"] == 0:                      # a certified child cannot sit below its parent (feasible sets nest, OMC.jl:2522)
            assert o["dual_bound"] >= root["dual_bound"] - 2e-6 * abs(root["objective"])
        else:                                            # SLOW_PROGRESS: the bound is valid but may be loose (gap reported)
            assert o["feasible"] and o["dual_bound"] <= root["objective"] * (1 + 1e-2)
        # the certificate is two-sided: an OPTIMAL objective sits within 1e-6 (relative) of the certified bound, on either side
        assert abs(o["dual_bound"] - o["objective"]) <= 1.01e-6 * abs(o["objective"]) or o["status_code"] != 0
        assert_finite(o)
    ev = eng.evaluate_objective(root["X"])
    assert ev >= root["objective"] - 1e-6 * abs(ev)      # the relaxation value is below the master objective of its own X
    eng.close()


def test_altmin_and_rounding_match_oracle(have_gpu, omc, orc):
    """alternating_minimization (OMC.jl:1979-2279) and the SVD rounding glue (OMC.jl:873), rank 1."""
    n, m, k = 30, 36, 1
    A, mask = orc.make_instance(n, m, k, seed=31, kind="lowrank", n_indices=int(0.3 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    eng = omc.Engine(A, mask, GAMMA, k)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 1)                       # OMC.jl:522-524
    rng = np.random.default_rng(1)
    x1 = np.linalg.qr(rng.standard_normal((n, 1)))[0][:, 0]; x2 = np.linalg.qr(rng.standard_normal((n, 1)))[0][:, 0]
    node_sets = [[], [(x1, U0 * 0.6, ["left"])], [(x1, U0 * 0.6, ["right"]), (x2, -U0 * 0.3, ["left"])]]
    starts = [U0, U0, U0 + 0.05 * rng.standard_normal((n, 1))]
    got = eng.alternating_minimization(starts, node_sets, "linear")
    for g, cuts, u0 in zip(got, node_sets, starts):
        r = orc.alternating_minimization(inst, u0, cuts, "linear")
        assert g["converged"] == r["converged"] and g["n_iters"] == r["n_iters"]
        assert np.allclose(g["objectives"], r["objectives"], rtol=1e-9)
        assert np.allclose(g["U"], r["U"], atol=1e-7) and np.allclose(g["V"], r["V"], atol=1e-6)
        assert np.linalg.norm(g["U"]) <= 1 + 1e-9 and g["U"][-1, 0] >= -1e-12
        X = g["U"] @ g["V"]
        assert eng.evaluate_objective(X) == pytest.approx(orc.evaluate_objective(X, A, mask, GAMMA), rel=1e-12)
    # rounding of a relaxed Y
    out = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=4.0))[0]
    Ur = eng.round_Y([out["Y"]])[0]
    ref = orc.svd_rounding(out["Y"], 1)
    assert np.allclose(Ur, ref, atol=1e-7)
    with pytest.raises(ValueError):
        eng.alternating_minimization([U0], [[]], "cubic")
    eng.close()


def test_continuous_batching_slots(have_gpu, omc, orc):
    """More nodes than slots: a finished slot is harvested and re-used; per-node results must not depend on the slot count."""
    n, m, k = 20, 24, 1
    A, mask = orc.make_instance(n, m, k, seed=41, kind="readme")
    inst = orc.Instance(A, mask, GAMMA, k)
    nodes = oracle_path(orc, inst, "linear", 5, 16.0, seed=9)
    nodes = nodes + nodes[1:4]                                              # 9 nodes
    eng = omc.Engine(A, mask, GAMMA, k)
    ref = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=omc.default_params(rho_scale=16.0))
    for slots in (1, 2, 4):
        out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=omc.default_params(rho_scale=16.0, slots=slots))
        for a, b in zip(out, ref):
            assert a["status_code"] == b["status_code"] and a["iters"] == b["iters"]
            assert a["objective"] == pytest.approx(b["objective"], rel=1e-12) and a["dual_bound"] == pytest.approx(b["dual_bound"], rel=1e-12)
            assert np.allclose(a["U"], b["U"], atol=1e-12) and np.allclose(a["X"], b["X"], atol=1e-12)
            assert np.allclose(a["breakpoint_vec"], b["breakpoint_vec"], atol=1e-10)
    eng.close()


def test_dense_columns_and_deep_paths(have_gpu, omc, orc):
    """Paths that the small cases above do not reach: columns with more than 64 observed rows (vectors no longer fit one
    lane each: LDS/global-scratch solves), and a deep node (12 cuts: 37 rows, order-14 small cone)."""
    rng = np.random.default_rng(3)
    n, m, k = 70, 72, 1
    A = rng.standard_normal((n, 1)) @ rng.standard_normal((1, m)) + 0.05 * rng.standard_normal((n, m))
    mask = rng.random((n, m)) < 0.97                                       # c ~ 68 per column
    inst = orc.Instance(A, mask, GAMMA, k)
    assert max(len(c) for c in inst.cols) > 64
    eng = omc.Engine(A, mask, GAMMA, k)
    g = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=8.0))[0]
    r = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=8.0), want_certificate=False)
    assert g["status_code"] == r["termination_status"] and g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    eng.close()
    # columns with 41..64 observed rows: the layout without a copy of B (a second factorization gathers B again)
    n, m, k = 56, 60, 1
    A = rng.standard_normal((n, 1)) @ rng.standard_normal((1, m)) + 0.05 * rng.standard_normal((n, m))
    mask = rng.random((n, m)) < 0.9
    inst = orc.Instance(A, mask, GAMMA, k)
    assert 40 < max(len(c) for c in inst.cols) <= 64
    eng = omc.Engine(A, mask, GAMMA, k)
    g = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=8.0))[0]
    r = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=8.0), want_certificate=False)
    assert g["status_code"] == r["termination_status"] == 0 and g["iters"] == r["iters"]
    assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    eng.close()
    # deep path
    n, m = 20, 24
    A, mask = orc.make_instance(n, m, 1, seed=51, kind="readme")
    inst = orc.Instance(A, mask, GAMMA, 1)
    nodes = oracle_path(orc, inst, "linear", 12, 16.0, seed=2)
    eng = omc.Engine(A, mask, GAMMA, 1)
    sel = [nodes[4], nodes[8], nodes[12]]
    out = eng.matrix_completion_SDP_relaxation(sel, "linear", params=omc.default_params(rho_scale=16.0))
    for g, c in zip(out, sel):
        r = orc.sdp_relaxation(inst, c, "linear", params=orc.RelaxParams(rho_scale=16.0), want_certificate=False)
        assert g["status_code"] == r["termination_status"]
        if g["status_code"] != 3:
            assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL) and g["iters"] == r["iters"]
    eng.close()


def test_more_than_sixteen_cut_rows(have_gpu, omc, orc):
    """k_global forms the quadratic forms x' tY x of the cut rows 16 at a time on the matrix cores and stages 16 active cut vectors in LDS:
    nodes with 18 and 21 cuts take the second pass and the beyond-16 path of the final update."""
    n, m = 18, 22
    A, mask = orc.make_instance(n, m, 1, seed=77, kind="readme")
    inst = orc.Instance(A, mask, GAMMA, 1)
    nodes = oracle_path(orc, inst, "linear", 21, 16.0, seed=5)
    eng = omc.Engine(A, mask, GAMMA, 1)
    sel = [nodes[16], nodes[18], nodes[21]]
    out = eng.matrix_completion_SDP_relaxation(sel, "linear", params=omc.default_params(rho_scale=16.0))
    for g, c in zip(out, sel):
        r = orc.sdp_relaxation(inst, c, "linear", params=orc.RelaxParams(rho_scale=16.0), want_certificate=False)
        assert_finite(g)
        assert g["status_code"] == r["termination_status"], (len(c), g["status_code"], r["termination_status"])
        if g["status_code"] != 3:
            assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
            assert abs(g["iters"] - r["iters"]) <= 25, (len(c), g["iters"], r["iters"])
    eng.close()


def test_large_order_uses_l2_resident_path(have_gpu, omc, orc):
    """n = 150: G of the cone kernel (150 x 162 x 8 B) does not fit the LDS budget -> L2-resident variant of that kernel; the packed
    lower triangle of k_global's target (91 KB) still does (its L2-resident variant runs at n = 200, test_order_200_l2_resident_variants_agree).
    Oracle comparison on the root (the oracle needs ~1 min here)."""
    n, m, k = 150, 150, 1
    A, mask = orc.make_instance(n, m, k, seed=61, kind="lowrank", n_indices=int(0.15 * n * m))
    eng = omc.Engine(A, mask, GAMMA, k)
    P = omc.default_params(rho_scale=4.0, max_iters=400)
    g = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P)[0]
    info = eng.solver_info()
    assert not info["cone_lds"] and info["global_lds"]
    inst = orc.Instance(A, mask, GAMMA, k)
    r = orc.sdp_relaxation(inst, params=orc.RelaxParams(rho_scale=4.0, max_iters=400), want_certificate=False)
    assert g["iters"] == r["iters"] and g["status_code"] == r["termination_status"]
    assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    assert g["dual_bound"] == pytest.approx(r["dual_bound"], rel=1e-5)
    eng.tuning_set("OMC_GLOBAL_NOLDS", "1")         # the same solve with the target of k_global in its L2-resident scratch
    try:
        g2 = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P)[0]
        assert not eng.solver_info()["global_lds"]
    finally:
        eng.tuning_set("OMC_GLOBAL_NOLDS", None)
    assert g2["iters"] == g["iters"] and g2["status_code"] == g["status_code"]
    assert g2["objective"] == pytest.approx(g["objective"], rel=1e-10) and g2["dual_bound"] == pytest.approx(g["dual_bound"], rel=1e-8)
    eng.close()


# ---- Shor minors (rows a10 / a11): integer work, bit-exact against the oracle ----------------------------------------
def _shor_instance(orc, omc, n, m, k, frac, seed):
    rng = np.random.default_rng(seed)
    mask = rng.random((n, m)) < frac
    mask[rng.integers(0, n, m), np.arange(m)] = True
    mask[np.arange(n), rng.integers(0, m, n)] = True
    A = rng.standard_normal((n, m))
    return A, mask, omc.Engine(A, mask, GAMMA, k)


@pytest.mark.parametrize("n,m,frac,seed", [(5, 6, 0.5, 0), (7, 9, 0.4, 1), (12, 20, 0.3, 2), (9, 70, 0.5, 3), (20, 130, 0.2, 4)])
def test_shor_indexes_bit_exact(have_gpu, omc, orc, n, m, frac, seed):
    A, mask, eng = _shor_instance(orc, omc, n, m, 1, frac, seed)
    for cl in ([4], [3], [2], [1], [0], [4, 3, 2, 1, 0], [2, 4], [4, 4], [7], []):
        if n * m > 1500 and (0 in cl or 1 in cl or 2 in cl) and len(cl) > 1:
            continue                                   # keep the pure-Python oracle in seconds
        want = np.array(orc.shor_constraints_indexes(mask, cl), dtype=np.int64).reshape(-1, 4)
        got = eng.generate_rank1_matrix_completion_Shor_constraints_indexes(cl)
        assert got.shape == want.shape, (cl, got.shape, want.shape)
        assert np.array_equal(got, want), cl
        assert np.array_equal(eng.shor_count(cl), [len(orc.shor_constraints_indexes(mask, [p])) for p in cl])
    eng.close()


@pytest.mark.parametrize("k", [1, 2])
def test_violated_shor_minors_bit_exact(have_gpu, omc, orc, k):
    n, m = 10, 14
    A, mask, eng = _shor_instance(orc, omc, n, m, k, 0.45, 10 + k)
    rng = np.random.default_rng(5)
    cases = {"gauss": rng.standard_normal((k, n, m)), "ties": rng.integers(-2, 3, (k, n, m)).astype(float), "zero": np.zeros((k, n, m))}
    for name, X3 in cases.items():
        for cl in ([4], [4, 3], [3, 2, 1, 0, 4]):
            first = orc.violated_shor_minors(X3, mask, cl, [], 7)
            existing = [t for _, t in first[:5]] + [(1, 2, 1, 2), (n, n, m, m)]
            for ex, nm in (([], 7), (existing, 100), (existing, 10 ** 6), ([], 0)):
                want = orc.violated_shor_minors(X3, mask, cl, ex, nm)
                got = eng.generate_violated_Shor_minors(X3, cl, ex, nm)
                assert len(got) == len(want), (name, cl, nm)
                assert [t for _, t in got] == [t for _, t in want], (name, cl, nm)
                assert [s for s, _ in got] == [s for s, _ in want], (name, cl, nm)      # identical doubles
    eng.close()


def test_shor_counts_config3_size(have_gpu, omc, orc):
    """200 x 200, 20 % observed (BASELINE config 3): counts against popcount formulas, order and uniqueness properties."""
    A, mask, gamma, c = omc.pkg.data.config_instance(3, seed=0)
    eng = omc.Engine(A, mask, gamma, c["k"])
    Mi = mask.astype(np.int64)
    both = Mi @ Mi.T; obs = Mi.sum(1); m = mask.shape[1]
    xor = obs[:, None] + obs[None, :] - 2 * both; none = m - obs[:, None] - obs[None, :] + both
    iu = np.triu_indices(mask.shape[0], 1)
    b, x, z = both[iu], xor[iu], none[iu]
    want = {4: (b * (b - 1) // 2).sum(), 3: (b * x).sum(), 2: (b * z).sum() + (x * (x - 1) // 2).sum(), 1: (x * z).sum(), 0: (z * (z - 1) // 2).sum()}
    got = eng.shor_count([4, 3, 2, 1, 0])
    assert [int(v) for v in got] == [int(want[p]) for p in (4, 3, 2, 1, 0)]
    T = eng.generate_rank1_matrix_completion_Shor_constraints_indexes([4])
    assert len(T) == want[4]
    assert (T[:, 0] < T[:, 1]).all() and (T[:, 2] < T[:, 3]).all()
    assert mask[T[:, 0] - 1, T[:, 2] - 1].all() and mask[T[:, 0] - 1, T[:, 3] - 1].all() and mask[T[:, 1] - 1, T[:, 2] - 1].all() and mask[T[:, 1] - 1, T[:, 3] - 1].all()
    key = ((T[:, 0] * 1000 + T[:, 1]) * 1000 + T[:, 2]) * 1000 + T[:, 3]
    assert (np.diff(key) > 0).all()                     # the push order of p = 4 is lexicographic, hence also unique
    # top-100 violated minors of a rank-1 + noise X: scores must be the 100 largest, in order
    rng = np.random.default_rng(0)
    X3 = (A + 0.1 * rng.standard_normal(A.shape))[None]
    top = eng.generate_violated_Shor_minors(X3, [4], [], 100)
    sc = np.abs(X3[0][T[:, 0] - 1, T[:, 2] - 1] * X3[0][T[:, 1] - 1, T[:, 3] - 1] - X3[0][T[:, 0] - 1, T[:, 3] - 1] * X3[0][T[:, 1] - 1, T[:, 2] - 1])
    order = np.lexsort((key, sc))[::-1][:100]
    assert [t for _, t in top] == [tuple(int(v) for v in T[i]) for i in order]
    assert [s for s, _ in top] == [float(sc[i]) for i in order]
    eng.close()


def test_anderson_acceleration_option(have_gpu, omc, orc):
    """accel = 1 (k_aa): same certified optimum as the plain iteration, far fewer iterations on the crawling node of the
    README fixture (node L = 2: SLOW_PROGRESS at 3000 iterations without, OPTIMAL with), and the same trajectory as the
    oracle's mirror of the scheme on these small cases."""
    z = np.load(os.path.join(HERE, "golden", "readme_20x24_k1_linear.npz"), allow_pickle=False)
    DN = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
    cuts = [(z["cut_x"][l], z["cut_U"][l], [DN[int(c)] for c in z["cut_dir"][l]]) for l in range(len(z["cut_x"]))]
    nodes = [cuts[:int(L)] for L in z["node_L"]]
    eng = omc.Engine(z["A"], z["mask"], 80.0, 1)
    plain = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=omc.default_params(rho_scale=16.0, accel=0))
    fast = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=omc.default_params(rho_scale=16.0, accel=1))
    inst = orc.Instance(z["A"], z["mask"], 80.0, 1)
    assert plain[2]["status_code"] == 1 and fast[2]["status_code"] == 0 and fast[2]["iters"] < 500
    for b, (p_, f_) in enumerate(zip(plain, fast)):
        assert f_["status_code"] == 0 and f_["iters"] <= p_["iters"]
        assert f_["objective"] == pytest.approx(p_["objective"], rel=OBJ_REL)
        assert abs(f_["objective"] - f_["dual_bound"]) <= 1.01e-6 * max(1.0, abs(f_["objective"]))
        assert f_["dual_bound"] <= p_["objective"] * (1 + 1e-7)
        r = orc.sdp_relaxation(inst, nodes[b], "linear", params=orc.RelaxParams(rho_scale=16.0, accel=1), want_certificate=False)
        assert r["termination_status"] == 0 and abs(r["iters"] - f_["iters"]) <= 25
        assert f_["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    eng.close()


def test_branch_and_bound_rank2_smoke(have_gpu, omc, orc):
    """Rank-2 tree (4 children per node, GPU altmin for k = 2): counters and bounds stay consistent."""
    A, mask = orc.make_instance(16, 20, 2, seed=13, kind="lowrank", n_indices=int(0.4 * 16 * 20))
    eng = omc.Engine(A, mask, GAMMA, 2)
    sol, inst = omc.pkg.bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=60.0, batch=16, rho_scale=4.0, use_max_steps=True, max_steps=60)
    c = inst["run_details"]; log = np.array(inst["run_log"])
    assert c["nodes_dominated"] + c["nodes_relax_infeasible"] + c["nodes_relax_feasible"] == c["nodes_explored"]
    assert (np.diff(log[:, 3]) >= -1e-9).all() and sol["lower_bound"] <= sol["objective"] * (1 + 1e-9)
    assert np.linalg.matrix_rank(sol["X"], tol=1e-8) <= 2
    assert sol["objective"] == pytest.approx(orc.evaluate_objective(sol["X"], A, mask, GAMMA), rel=1e-10)
    eng.close()


def test_branch_and_bound_invariants(have_gpu, omc, orc):
    """Driver counterpart (bnb.branch_and_bound) on the README-type 20 x 24 instance: the per-run invariants of SURVEY 8c --
    LB monotone, LB <= UB, counter identity (OMC.jl:411-425), incumbent = evaluate_objective(X) of a rank-k X, and a root
    bound that the oracle confirms."""
    A, mask = orc.make_instance(20, 24, 1, seed=11, kind="readme")
    eng = omc.Engine(A, mask, GAMMA, 1)
    sol, inst = omc.pkg.bnb.branch_and_bound(eng, A, mask, gap=1e-3, time_limit=60.0, batch=16, rho_scale=16.0, use_max_steps=True, max_steps=150,
                                             altmin_root_n_iters=3)
    log = np.array(inst["run_log"]); c = inst["run_details"]
    assert len(log) >= 2
    lbs, ubs = log[:, 3], log[:, 4]
    assert (np.diff(lbs) >= -1e-9).all() and (np.diff(ubs) <= 1e-12).all()            # OMC.jl:1213-1216; incumbent only improves
    assert (lbs <= ubs * (1 + 1e-9)).all() and sol["lower_bound"] <= sol["objective"] * (1 + 1e-9)
    assert c["nodes_dominated"] + c["nodes_relax_infeasible"] + c["nodes_relax_feasible"] == c["nodes_explored"]
    assert c["nodes_relax_feasible"] >= c["nodes_relax_feasible_pruned"] + c["nodes_master_feasible"] + c["nodes_relax_feasible_split"]
    assert np.linalg.matrix_rank(sol["X"], tol=1e-8) <= 1
    assert sol["objective"] == pytest.approx(orc.evaluate_objective(sol["X"], A, mask, GAMMA), rel=1e-10)
    inst_o = orc.Instance(A, mask, GAMMA, 1)
    root = orc.sdp_relaxation(inst_o, params=orc.RelaxParams(rho_scale=16.0), want_certificate=False)
    assert log[0, 3] == pytest.approx(root["dual_bound"], rel=2e-6)                   # first logged LB = certified root bound
    assert root["dual_bound"] <= sol["objective"] * (1 + 1e-9)
    eng.close()


@pytest.mark.parametrize("k,cut_type", [(2, "linear"), (2, "linear3"), (3, "linear2")])
def test_altmin_rank_k_matches_oracle(have_gpu, omc, orc, k, cut_type):
    """alternating_minimization for k > 1 (k_altmin_k): k^2 quadratic constraints by projected Newton on the dual, mirrored by the
    oracle's _ustep_dual_newton (itself validated against SLSQP in the CPU suite)."""
    n, m = 16, 22
    A, mask = orc.make_instance(n, m, k, seed=70 + k, kind="lowrank", n_indices=int(0.5 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    eng = omc.Engine(A, mask, GAMMA, k)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), k)
    rng = np.random.default_rng(2)
    dirs = orc.child_directions(cut_type, k)
    x1 = np.linalg.qr(rng.standard_normal((n, 1)))[0][:, 0]; x2 = np.linalg.qr(rng.standard_normal((n, 1)))[0][:, 0]
    node_sets = [[], [(x1, U0 * 0.6, list(dirs[1]))], [(x1, U0 * 0.6, list(dirs[-1])), (x2, -U0 * 0.3, list(dirs[0]))]]
    starts = [U0, U0, U0 + 0.05 * rng.standard_normal((n, k))]
    got = eng.alternating_minimization(starts, node_sets, cut_type, max_iters=40)
    W, rad = orc.quadratic_constraint_vectors(k)
    for g, cuts, u0 in zip(got, node_sets, starts):
        r = orc.alternating_minimization(inst, u0, cuts, cut_type, max_iters=40)
        assert g["converged"] == r["converged"] and g["n_iters"] == r["n_iters"]
        assert np.allclose(g["objectives"], r["objectives"], rtol=1e-8)
        assert np.allclose(g["U"] @ g["V"], r["U"] @ r["V"], atol=1e-5)
        assert (((g["U"] @ W.T) ** 2).sum(0) - rad).max() <= 1e-9                       # balls and pair cones (OMC.jl:2029-2045, 2164-2171)
        for j in range(k):
            assert (g["U"][n - k + j:, j] >= -1e-10).all()                               # symmetry breaking (OMC.jl:1989-1996)
        X = g["U"] @ g["V"]
        assert g["objectives"][-1] == pytest.approx(orc.evaluate_objective(X, A, mask, GAMMA), rel=1e-9)   # model_U objective = master objective at (U, V)
    eng.close()


def test_time_limit_and_iteration_cap_statuses(have_gpu, omc, orc):
    """MOI.TIME_LIMIT / SLOW_PROGRESS branches of the driver (OMC.jl:783-784, 841): values stay available, bounds stay valid."""
    A, mask = orc.make_instance(24, 28, 1, seed=12, kind="lowrank", n_indices=int(0.4 * 24 * 28))
    eng = omc.Engine(A, mask, GAMMA, 1)
    full = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=4.0))[0]
    assert full["status_code"] == 0
    capped = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=4.0, max_iters=50))[0]
    assert capped["termination_status"] == "SLOW_PROGRESS" and capped["feasible"] and capped["iters"] == 50
    assert capped["dual_bound"] <= full["objective"] * (1 + 1e-9)                       # still a valid lower bound
    # first_wins (penalty autotune): the batch ends at the first check with a certified node; the others come back as they stand
    race = eng.matrix_completion_SDP_relaxation([[], [], []], "linear", params=omc.default_params(rho_scale=1.0, first_wins=1), rho_scales=[0.05, 4.0, 300.0])
    assert race[1]["status_code"] == 0 and race[1]["iters"] == full["iters"]
    for o in (race[0], race[2]):
        assert o["status_code"] in (0, 1) and o["iters"] <= race[1]["iters"] and o["feasible"]
        assert o["dual_bound"] <= full["objective"] * (1 + 1e-9)
    assert race[0]["status_code"] == 1 or race[2]["status_code"] == 1
    timed = eng.matrix_completion_SDP_relaxation([[], []], "linear", params=omc.default_params(rho_scale=4.0, time_limit=1e-9))
    for o in timed:
        assert o["termination_status"] == "TIME_LIMIT" and o["feasible"]
        assert o["dual_bound"] <= full["objective"] * (1 + 1e-9)
    eng.close()


def test_order_200_l2_resident_variants_agree(have_gpu, omc):
    """BASELINE config 3 size (200 x 200): the 1024-thread launch of the L2-resident cone kernel against its 512-thread form
    (same arithmetic, different lane mapping), plus the certificate properties of the truncated solve."""
    A, mask, gamma, c = omc.pkg.data.config_instance(3, seed=0)
    eng = omc.Engine(A, mask, gamma, c["k"])
    P = omc.default_params(rho_scale=4.0, max_iters=125)
    a = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P, want_X=False)[0]
    eng.tuning_set("OMC_CONE_512", "1")
    try:
        b = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P, want_X=False)[0]
    finally:
        eng.tuning_set("OMC_CONE_512", None)
    assert a["iters"] == b["iters"] and a["status_code"] == b["status_code"]
    assert a["objective"] == pytest.approx(b["objective"], rel=1e-10) and a["dual_bound"] == pytest.approx(b["dual_bound"], rel=1e-8)
    assert a["dual_bound"] <= a["objective"] * (1 + 1e-6)
    w = np.linalg.eigvalsh(a["Y"])
    assert w[0] >= -1e-5 and w[-1] <= 1 + 1e-5
    eng.close()


def test_run_to_run_determinism(have_gpu, omc, orc):
    """No floating-point atomics on the path: two runs of the same batch return bit-identical objectives, bounds and iteration counts."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, c["k"])
    P = omc.default_params(rho_scale=4.0, max_iters=400)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 4, "linear", params=P)
    runs = []
    for _ in range(2):
        out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False)
        runs.append((np.array([o["objective"] for o in out]), np.array([o["dual_bound"] for o in out]), np.array([o["iters"] for o in out]),
                     np.stack([o["Y"] for o in out])))
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)
    eng.close()


# ---- round 2: rank 2 / linear3 / smallest_2_eigvec, the NNQP cap, altmin failure, config-5-size Shor selection ------------------
@pytest.mark.parametrize("q1", [True, False])
def test_rank2_linear3_smallest2_matches_oracle(have_gpu, omc, orc, q1):
    """BASELINE config 4's combination (rank 2, linear3 cuts, smallest_2_eigvec breakpoints; OMC.jl:1636-1678, 2469-2477) at 26 x 32,
    depth 3, with the reference's quirk Q1 (the `right` piece a*v, OMC.jl:1675) and with the secant it was meant to be.  The
    `right` piece under Q1 pins v_j = a (no Slater point): such children are drawn only in the Q1 run's last level."""
    n, m, k = 26, 32, 2
    A, mask = orc.make_instance(n, m, k, seed=91, kind="lowrank", n_indices=int(0.4 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    nodes = oracle_path(orc, inst, "linear3", 3, 4.0, seed=5, breakpoints="smallest_2_eigvec", q1=q1, avoid=("right",) if q1 else ())
    if q1:       # one pinned child (right / right) on top of the path, as a B&B run under Q1 creates them at every split
        nodes.append(nodes[2] + [(nodes[3][2][0], nodes[3][2][1], ["right", "inner_left"])])
    P = omc.default_params(rho_scale=4.0, breakpoints=2, reference_quirk_q1=int(q1))
    eng = omc.Engine(A, mask, GAMMA, k)
    out = eng.matrix_completion_SDP_relaxation(nodes, "linear3", params=P, want_Theta=True)
    ncert = 0
    for g, c in zip(out, nodes):
        r = orc.sdp_relaxation(inst, c, "linear3", params=orc.RelaxParams(rho_scale=4.0, reference_quirk_q1=q1))
        assert g["status_code"] == r["termination_status"], (len(c), g["termination_status"])
        assert_finite(g)
        if g["status_code"] == 3:
            continue
        assert g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
        x_ref, ev_ref = orc.breakpoint_vector(g["Y"], g["U"], "smallest_2_eigvec")       # OMC.jl:2469-2477 on the GPU's own (Y, U)
        assert g["lambda_min"][0] == pytest.approx(ev_ref[0], abs=1e-9)
        if ev_ref[1] - ev_ref[0] > 1e-6 and (len(ev_ref) < 3 or abs(ev_ref[1]) > 1e-8):
            assert np.allclose(g["breakpoint_vec"], x_ref, atol=1e-6) or np.allclose(g["breakpoint_vec"], -x_ref, atol=1e-6)
        if g["status_code"] != 0:
            continue
        ncert += 1
        assert abs(g["objective"] - g["dual_bound"]) <= 1.01e-6 * max(1.0, abs(g["objective"]))
        assert g["dual_bound"] == pytest.approx(r["dual_bound"], rel=OBJ_REL)
        Theta = 0.5 * (g["Theta"] + g["Theta"].T)
        res = orc.primal_residuals(inst, r["rows"], g["Y"], g["U"], g["X"], Theta)
        assert res["max"] <= 2e-5 * max(1.0, np.abs(Theta).max()), res
    assert ncert >= 2
    eng.close()


def test_nnqp_passive_set_cap_is_reported(have_gpu, omc, orc):
    """More simultaneously active rows than NNQP_PMAX (64): the row projection is no longer exact, which the device flags --
    such a node must not come back OPTIMAL; its bound stays valid.  70 box rows U_i <= 0.7 p_i all bind because the cut
    x = p (right piece) ties x'Yx to v = x'U and the objective wants x'Yx large."""
    n, m, k = 70, 72, 1
    rng = np.random.default_rng(8)
    p = np.abs(rng.standard_normal(n)) + 0.5; p /= np.linalg.norm(p)
    A = np.outer(3.0 * p, rng.standard_normal(m)) + 0.01 * rng.standard_normal((n, m))
    mask = rng.random((n, m)) < 0.3
    mask[rng.integers(0, n, m), np.arange(m)] = True; mask[np.arange(n), rng.integers(0, m, n)] = True
    Uhat = (0.3 * p)[:, None]                                   # v-hat = 0.3
    node = [(p, Uhat, ["right"])]
    U_upper = (0.7 * p)[:, None]
    eng = omc.Engine(A, mask, GAMMA, k)
    P = omc.default_params(rho_scale=4.0, max_iters=1500)
    g = eng.matrix_completion_SDP_relaxation([node], "linear", params=P, U_upper=[U_upper])[0]
    inst = orc.Instance(A, mask, GAMMA, k)
    r = orc.sdp_relaxation(inst, node, "linear", U_upper=U_upper, params=orc.RelaxParams(rho_scale=4.0, max_iters=1500), want_certificate=False)
    nact = int((r["lam"] > 0).sum())
    assert_finite(g)
    assert g["dual_bound"] <= r["objective"] * (1 + 1e-6) + 1e-9           # whatever happened to the rows, the bound is valid
    if nact > 64:
        assert g["status_code"] != 0, "rows were deferred by the passive-set cap: the node cannot be certified"
    else:
        assert g["status_code"] == r["termination_status"] and g["objective"] == pytest.approx(r["objective"], rel=OBJ_REL)
    eng.close()


def test_altmin_infeasible_model_U_reports_failure(have_gpu, omc, orc):
    """A cut set that leaves model_U without a feasible point (recorded by the round-1 random sweep): the reference's loop ends
    with converged = false (OMC.jl:2231, 2263-2265); both sides stop in the first iteration and record no objective."""
    z = np.load(os.path.join(HERE, "golden", "altmin_infeasible_6x13_k2.npz"), allow_pickle=False)
    k = int(z["k"]); ct = str(z["ct"])
    DN = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
    cuts = [(z["cut_x"][l], z["cut_U"][l], [DN[int(c)] for c in z["cut_dir"][l]]) for l in range(len(z["cut_x"]))]
    inst = orc.Instance(z["A"], z["mask"], GAMMA, k)
    eng = omc.Engine(z["A"], z["mask"], GAMMA, k)
    r = orc.alternating_minimization(inst, z["U0"], cuts, ct, max_iters=30)
    g = eng.alternating_minimization([z["U0"], z["U0"]], [cuts, []], ct, max_iters=30)
    assert not r["converged"] and r["n_iters"] == 1 and len(r["objectives"]) == 0
    assert not g[0]["converged"] and g[0]["n_iters"] == 1
    r2 = orc.alternating_minimization(inst, z["U0"], [], ct, max_iters=30)              # the same start without the cuts is fine
    assert g[1]["converged"] == r2["converged"] and g[1]["n_iters"] == r2["n_iters"]
    # rank 1: contradictory bounds on one direction
    n, m = 12, 15
    A, mask = orc.make_instance(n, m, 1, seed=4, kind="lowrank", n_indices=int(0.5 * n * m))
    x = np.zeros(n); x[0] = 1.0
    Uh = np.zeros((n, 1)); Uh[0, 0] = 0.5
    bad = [(x, Uh, ["right"]), (x, -Uh, ["left"])]            # 0.5 <= u_0 and u_0 <= -0.5
    e1 = omc.Engine(A, mask, GAMMA, 1); i1 = orc.Instance(A, mask, GAMMA, 1)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 1)
    g1 = e1.alternating_minimization([U0], [bad], "linear")[0]
    r1 = orc.alternating_minimization(i1, U0, bad, "linear")
    assert not r1["converged"] and not g1["converged"] and g1["n_iters"] == r1["n_iters"] == 1
    eng.close(); e1.close()


def test_violated_shor_minors_config5_size(have_gpu, omc):
    """BASELINE config 5 shape (1000 x 1000, k = 2, 30 % observed): ~2e9 four-present candidate minors are scored and the 100
    most violated selected on the device.  The reference materialises every candidate (OMC.jl:2621-2633), which no host can
    do here, so the check is by properties: scores recomputed for the returned tuples, order, validity, and no better minor
    among 2e6 randomly drawn candidates."""
    A, mask, gamma, c = omc.pkg.data.config_instance(5, seed=0)
    n, m = mask.shape; k = c["k"]
    eng = omc.Engine(A, mask, gamma, k)
    rng = np.random.default_rng(1)
    L = rng.standard_normal((k, n, 1)); R = rng.standard_normal((k, 1, m))
    X3 = L * R + 0.05 * rng.standard_normal((k, n, m))                      # k rank-1 slices + noise (OMC.jl:2627-2631 sums over t)
    top = eng.generate_violated_Shor_minors(X3, [4], [], 100)
    assert len(top) == 100
    def score(i1, i2, j1, j2):
        s = 0.0
        for t in range(k):
            s += abs(X3[t, i1, j1] * X3[t, i2, j2] - X3[t, i1, j2] * X3[t, i2, j1])
        return s
    sc = [s for s, _ in top]
    assert all(a > b or (a == b) for a, b in zip(sc, sc[1:]))
    keys = [(s, t) for s, t in top]
    assert keys == sorted(keys, reverse=True) and len(set(t for _, t in top)) == 100
    for s, (i1, i2, j1, j2) in top:
        assert 1 <= i1 < i2 <= n and 1 <= j1 < j2 <= m
        assert mask[i1 - 1, j1 - 1] and mask[i1 - 1, j2 - 1] and mask[i2 - 1, j1 - 1] and mask[i2 - 1, j2 - 1]
        assert s == score(i1 - 1, i2 - 1, j1 - 1, j2 - 1)                    # identical doubles (no FMA contraction in the scorer)
    # random four-present candidates: none beats the 100th score
    i1 = rng.integers(0, n, 4_000_000); i2 = rng.integers(0, n, 4_000_000); j1 = rng.integers(0, m, 4_000_000); j2 = rng.integers(0, m, 4_000_000)
    ok = (i1 < i2) & (j1 < j2)
    i1, i2, j1, j2 = i1[ok], i2[ok], j1[ok], j2[ok]
    ok = mask[i1, j1] & mask[i1, j2] & mask[i2, j1] & mask[i2, j2]
    i1, i2, j1, j2 = i1[ok], i2[ok], j1[ok], j2[ok]
    s = np.zeros(len(i1))
    for t in range(k):
        s += np.abs(X3[t, i1, j1] * X3[t, i2, j2] - X3[t, i1, j2] * X3[t, i2, j1])
    assert len(s) > 1000 and s.max() <= sc[-1] or s.max() in sc
    st = eng.shor_last_stats()
    assert st["candidates"] > 1_000_000_000
    eng.close()


def test_config4_shape_capped_iterations_properties(have_gpu, omc):
    """BASELINE config 4 shape (500 x 500, rank 2, linear3 + smallest_2_eigvec): the oracle would need hours, so a capped solve is checked
    by properties -- finite outputs, a valid bound below the primal value of the (nearly feasible) iterate, Y inside the cone to the
    accuracy the residual reports, and the cone block really running on the tracked subspace (no eigendecomposition after the seed)."""
    A, mask, gamma, c = omc.pkg.data.config_instance(4, seed=0)
    eng = omc.Engine(A, mask, gamma, c["k"])
    P = omc.default_params(rho_scale=4.0, max_iters=100, breakpoints=2)
    root = eng.matrix_completion_SDP_relaxation([[]], "linear3", params=P, want_X=False)[0]
    assert_finite(root)
    assert root["iters"] == 100 and root["feasible"]
    assert root["dual_bound"] <= root["objective"] * (1 + 1e-3)            # a valid bound under the value of a nearly feasible point
    w = np.linalg.eigvalsh(root["Y"])
    assert w[0] >= -1e-3 and w[-1] <= 1 + 1e-3 and np.trace(root["Y"]) <= 2 + 1e-6
    st = eng.subspace_stats()
    assert st["calls"] >= 60 and st["fallbacks"] <= 5, st                 # the order-500 projection is done by the 16-vector block
    x = root["breakpoint_vec"]
    assert abs(np.linalg.norm(x) - 1.0) < 1e-9                            # smallest_2 mix has unit norm (w1^2 + w2^2 = 1, OMC.jl:2471-2476)
    # one level of children (16 = 4^2 direction pairs, OMC.jl:2481-2491) is a legal batch at this size
    kids = omc.pkg.bnb.make_children([], root, "linear3", 2)
    assert len(kids) == 16
    out = eng.matrix_completion_SDP_relaxation(kids[:4], "linear3", params=omc.default_params(rho_scale=4.0, max_iters=50, breakpoints=2), want_X=False, want_Y=False)
    for o in out:
        assert_finite(o)
        assert o["status_code"] in (0, 1, 3)
    eng.close()


def test_altmin_config2_size_deep_cut_list(have_gpu, omc, orc):
    """alternating_minimization at BASELINE config 2 size (100 x 100, rank 1) with a depth-8 cut list (16 bound rows + symmetry row):
    same iteration count and objectives as the oracle, and the device-side master objective of U V (OMC.jl:920-927) equals
    evaluate_objective of the product."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=1)
    n, m = A.shape
    inst = orc.Instance(A, mask, gamma, 1)
    eng = omc.Engine(A, mask, gamma, 1)
    rng = np.random.default_rng(4)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 1)
    cuts = []
    for l in range(8):
        x = rng.standard_normal(n); x /= np.linalg.norm(x)
        vhat = float(U0[:, 0] @ x)
        cuts.append((x, U0 * (0.8 if l % 2 else 1.2), ["left" if (vhat * (0.8 if l % 2 else 1.2)) > vhat else "right"]))   # every cut keeps U0 feasible
    got = eng.alternating_minimization([U0, U0 + 0.02 * rng.standard_normal((n, 1))], [cuts, cuts], "linear")
    for g, u0 in zip(got, [U0, None]):
        if u0 is None:
            continue
        r = orc.alternating_minimization(inst, u0, cuts, "linear")
        assert g["converged"] == r["converged"] and g["n_iters"] == r["n_iters"]
        assert np.allclose(g["objectives"], r["objectives"], rtol=1e-9)
        assert np.allclose(g["U"], r["U"], atol=1e-7)
    for g in got:
        X = g["U"] @ g["V"]
        assert g["master_objective"] == pytest.approx(orc.evaluate_objective(X, A, mask, gamma), rel=1e-12)
        assert g["master_objective"] == pytest.approx(eng.evaluate_objective(X), rel=1e-12)
    eng.close()


def test_rccl_communicator_single_rank(have_gpu, omc):
    """omc_comm_init / omc_allreduce_bounds / omc_bcast_incumbent (C ABI over RCCL) with a world of one rank: the
    collectives run through librccl on the device and return the inputs."""
    A, mask, gamma, c = omc.pkg.data.config_instance(1, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    eng.comm_init(0, 1, omc.Engine.comm_unique_id())
    ub, lb, owner = eng.allreduce_bounds(3.5, -1.25)
    assert (ub, lb, owner) == (3.5, -1.25, 0)
    X = np.arange(A.size, dtype=float).reshape(A.shape)
    assert np.array_equal(eng.bcast_incumbent(0, X), X)
    rows = np.arange(21, dtype=float).reshape(3, 7)                        # omc_allgather_records: the per-node records of every rank
    assert np.array_equal(eng.allgather_records(rows, 7, 16), rows)
    assert eng.allgather_records(np.zeros((0, 7)), 7, 16).shape == (0, 7)
    with pytest.raises(omc.OmcError):
        eng.allgather_records(rows, 7, 2)                                 # out too small for the gathered rows
    # the driver counterpart takes every exchange through the engine's communicator once it exists
    comm = omc.pkg.bnb.Comm(0, 1, eng)
    assert np.array_equal(comm.allgather_rows(rows, 7), rows)
    with pytest.raises(omc.OmcError):
        eng.comm_init(0, 1, omc.Engine.comm_unique_id())                  # already initialised
    eng.close()


def test_asynchronous_submit_poll_wait(have_gpu, omc, orc):
    """omc_relax_submit / poll / wait: the solve runs on the library's worker thread while the host keeps working; results are those of
    the blocking call."""
    import time
    A, mask = orc.make_instance(24, 28, 1, seed=12, kind="lowrank", n_indices=int(0.4 * 24 * 28))
    inst = orc.Instance(A, mask, GAMMA, 1)
    nodes = oracle_path(orc, inst, "linear", 3, 4.0, seed=3)
    eng = omc.Engine(A, mask, GAMMA, 1)
    P = omc.default_params(rho_scale=4.0, slots=2)
    ref = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P)
    eng.stage(nodes, "linear", P)
    eng.submit()
    with pytest.raises(omc.OmcError):
        eng.submit()                                   # one solve in flight per handle
    seen = []; host_work = 0
    while True:
        st = eng.poll(); seen.append(st["nodes_done"])
        host_work += int(np.linalg.norm(np.ones(64)))  # the host thread is free
        if not st["running"]:
            break
        time.sleep(0.002)
    eng.wait()
    with pytest.raises(omc.OmcError):
        eng.wait()                                     # nothing in flight any more
    out = eng.fetch()
    assert seen == sorted(seen) and eng.poll()["nodes_done"] == len(nodes) == eng.poll()["nodes_total"]
    for a, b in zip(out, ref):
        assert a["status_code"] == b["status_code"] and a["iters"] == b["iters"]
        assert a["objective"] == b["objective"] and a["dual_bound"] == b["dual_bound"]
    eng.close()


# ---- round 3: tightened parity ("weak: parity") ---------------------------------------------------------------------------
def test_config2_full_size_depth3_cut_node_matches_oracle(have_gpu, omc, orc):
    """BASELINE config 2 at the size the metric is quoted on (100 x 100, rank 1, 20 % observed): a depth-3 cut node against the oracle at
    2e-6 (the oracle takes about a minute per solve here; the path is built from GPU results so that it is paid for once)."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    inst = orc.Instance(A, mask, gamma, 1)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0)
    cuts = []
    for d in range(3):
        r = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P, want_X=False)[0]
        assert r["feasible"]
        cuts = cuts + [(r["breakpoint_vec"].copy(), r["U"].copy(), ["left" if d % 2 == 0 else "right"])]
    g = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P)[0]
    o = orc.sdp_relaxation(inst, cuts, "linear", params=orc.RelaxParams(rho_scale=4.0))
    assert g["status_code"] == 0 and o["termination_status"] == orc.OMC_OPTIMAL
    assert g["objective"] == pytest.approx(o["objective"], rel=OBJ_REL)
    assert abs(g["objective"] - g["dual_bound"]) <= 1.01e-6 * abs(g["objective"])
    rows = orc.build_rows(inst, cuts, "linear")
    Theta = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P, want_Theta=True)[0]["Theta"]
    res = orc.primal_residuals(inst, rows, g["Y"], g["U"], g["X"], Theta)
    # the order-200 cone is dominated by Theta (||Theta||_2 ~ ||X||_F^2 ~ 1e4 here): its residual is judged relative to that
    assert res["psd_YXTheta"] <= 1e-8 * np.linalg.norm(Theta, 2), res
    assert max(v for kk, v in res.items() if kk not in ("max", "psd_YXTheta")) <= 2e-6, res
    eng.close()


def test_gpu_result_against_the_independent_slsqp_solve(have_gpu, omc, orc):
    """The independent check of tests/test_oracle_kats.py (SLSQP over Y = uu' + RR' with the cut rows of OMC.jl:1580-1683 written out by
    hand) applied to the GPU's result directly: every feasible value SLSQP reaches stays above the GPU's dual bound, and the converged
    value equals the GPU's objective -- no oracle in between."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(7)
    n, m, k = 5, 6, 1
    A, mask = orc.make_instance(n, m, k, seed=41, kind="lowrank", n_indices=18, noise=0.3)
    inst = orc.Instance(A, mask, GAMMA, k)
    eng = omc.Engine(A, mask, GAMMA, k)
    P = omc.default_params(rho_scale=8.0)
    root = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P)[0]
    x = root["breakpoint_vec"].copy()
    cut = (x, root["U"].copy(), ["right"])
    g = eng.matrix_completion_SDP_relaxation([[cut]], "linear", params=P)[0]
    assert g["status_code"] in (0, 1)
    vhat = float(root["U"][:, 0] @ x)
    lo, hi, sl, ic = orc.cut_piece("linear", "right", vhat)
    iu = np.tril_indices(n)

    def unpack(z):
        u = z[:n]; R = np.zeros((n, n)); R[iu] = z[n:]
        return u, np.outer(u, u) + R @ R.T

    cons = [
        dict(type="ineq", fun=lambda z: 1.0 - np.trace(unpack(z)[1])),
        dict(type="ineq", fun=lambda z: 1.0 - np.linalg.eigvalsh(unpack(z)[1])[-1]),
        dict(type="ineq", fun=lambda z: 1.0 - float(z[:n] @ z[:n])),
        dict(type="ineq", fun=lambda z: z[n - 1]),
        dict(type="ineq", fun=lambda z: float(x @ z[:n]) - lo), dict(type="ineq", fun=lambda z: hi - float(x @ z[:n])),
        dict(type="ineq", fun=lambda z: sl * float(x @ z[:n]) + ic - float(x @ unpack(z)[1] @ x)),
    ]
    best = np.inf
    for trial in range(6):
        u0 = g["U"][:, 0] + 0.05 * rng.standard_normal(n) if trial == 0 else 0.3 * rng.standard_normal(n)
        z0 = np.concatenate([u0, 0.2 * rng.standard_normal(len(iu[0]))])
        res = minimize(lambda z: inst.f_value(unpack(z)[1]), z0, constraints=cons, bounds=[(-1, 1)] * n + [(None, None)] * len(iu[0]), method="SLSQP",
                       options=dict(ftol=1e-13, maxiter=500))
        if all(cn["fun"](res.x) >= -1e-8 for cn in cons):
            assert g["dual_bound"] <= res.fun + 1e-7 * abs(res.fun)
            best = min(best, res.fun)
    assert np.isfinite(best)
    if g["status_code"] == 0:
        assert g["objective"] == pytest.approx(best, rel=2e-5)
    eng.close()


def test_separation_from_the_tracked_block_agrees_with_the_cold_eigendecomposition(have_gpu, omc, orc):
    """the separation vector of a harvested slot may come from the tracked block (k_cone_sub<2>), which only proves that its pair
    is AN eigenpair of Y - UU'.  On a pure-noise instance (README type: Y - UU' keeps many significant eigenvalues) and on a low-rank one,
    every node of a depth-4 frontier is checked against omc_separation_batch (cold eigendecomposition of the returned (Y, U)): same
    lambda_min, same vector up to the eigen-gap, and no node is called master-feasible by one path and not by the other."""
    for kind, n, m, nidx, rs in (("readme", 60, 64, None, 16.0), ("lowrank", 64, 72, 1400, 4.0)):
        A, mask = orc.make_instance(n, m, 1, n_indices=nidx, seed=3, noise=0.1, kind=kind)
        eng = omc.Engine(A, mask, GAMMA, 1)
        P = omc.default_params(rho_scale=rs, max_iters=1500)
        nodes, _ = omc.pkg.bnb.expand_frontier(eng, 4, "linear", params=P)
        out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False)
        ok = [o for o in out if o["feasible"]]
        xs, lam, _ = eng.breakpoint_vectors([o["Y"] for o in ok], [o["U"] for o in ok])
        for o, xc, lc in zip(ok, xs, lam):
            assert o["lambda_min"][0] == pytest.approx(lc[0], abs=1e-7)
            assert (o["lambda_min"][0] >= -1e-6) == (lc[0] >= -1e-6)
            if lc[1] - lc[0] > 1e-4:                       # a simple eigenvalue: the vector is determined up to its (canonical) sign
                assert np.allclose(o["breakpoint_vec"], xc, atol=1e-5)
        eng.close()


def test_stalled_nodes_certified_at_the_relaxed_residual_pass_the_primal_residuals(have_gpu, omc, orc):
    """a node whose values have been stationary for `stall_checks` checks is still reported OPTIMAL when its two-sided gap
    is closed and its cone residual is within 10x of eps_feas sqrt(n + k) (k_check_final).  Whatever path certified a node, the returned
    point must pass the reference's rows and cones at the tolerance the certificate claims."""
    A, mask = orc.make_instance(20, 24, 1, seed=11, kind="readme")
    inst = orc.Instance(A, mask, GAMMA, 1)
    eng = omc.Engine(A, mask, GAMMA, 1)
    P = omc.default_params(rho_scale=16.0, stall_checks=4)          # a short stall window makes the relaxed path fire
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 4, "linear", params=P)
    out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_Theta=True)
    nopt = 0
    for cuts, o in zip(nodes, out):
        if o["status_code"] != 0:
            continue
        nopt += 1
        rows = orc.build_rows(inst, cuts, "linear")
        res = orc.primal_residuals(inst, rows, o["Y"], o["U"], o["X"], o["Theta"])
        assert res["max"] <= 10 * 1e-7 * np.sqrt(21) * 1.5, res            # 10 x eps_feas sqrt(n + k), with slack for the recovery of U
        assert abs(o["objective"] - o["dual_bound"]) <= 1.01e-6 * max(1.0, abs(o["objective"]))
    assert nopt >= len(nodes) // 2
    eng.close()


# ---- round 3: warm start from the parent's final state (omc_state_pool_create / omc_relax_set_warm) ----------------------------------------
def test_warm_start_same_optimum_fewer_iterations_and_oracle_mirror(have_gpu, omc, orc):
    """A child is its parent plus one cut.  Warm-started from the parent's final state it must reach the same certified value as from a cold
    start (same convex program, 2e-6 on certified nodes), the oracle's warm start (same state transfer: Y, rescaled duals of the two full
    cones, column multipliers, U) must agree with the GPU's, and along a depth-4 path the warm solves need fewer iterations in total."""
    A, mask = orc.make_instance(24, 28, 1, seed=12, kind="lowrank", n_indices=int(0.4 * 24 * 28), noise=0.2)
    inst = orc.Instance(A, mask, GAMMA, 1)
    eng = omc.Engine(A, mask, GAMMA, 1)
    eng.state_pool_create(8)
    P = omc.default_params(rho_scale=4.0)
    OP = orc.RelaxParams(rho_scale=4.0)
    cuts = []; it_cold = it_warm = 0
    g = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P, save_to=[0])[0]
    o = orc.sdp_relaxation(inst, cuts, "linear", params=OP, want_certificate=False)
    for d in range(4):
        cuts = cuts + [(g["breakpoint_vec"].copy(), g["U"].copy(), ["left" if d % 2 == 0 else "right"])]
        cold = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P)[0]
        warm = eng.matrix_completion_SDP_relaxation([cuts], "linear", params=P, load_from=[d], save_to=[d + 1])[0]
        ow = orc.sdp_relaxation(inst, cuts, "linear", params=OP, want_certificate=False, warm=o["warm"])
        assert warm["status_code"] == 0 and ow["termination_status"] == orc.OMC_OPTIMAL
        assert warm["objective"] == pytest.approx(ow["objective"], rel=OBJ_REL)
        if cold["status_code"] == 0:
            assert warm["objective"] == pytest.approx(cold["objective"], rel=OBJ_REL)
        assert abs(warm["objective"] - warm["dual_bound"]) <= 1.01e-6 * max(1.0, abs(warm["objective"]))
        assert abs(warm["iters"] - ow["iters"]) <= 50                      # same transfer, same splitting (the tracked block makes projections inexact)
        it_cold += cold["iters"]; it_warm += warm["iters"]
        g = warm; o = ow
    assert it_warm < it_cold
    eng.close()


def test_warm_start_mixed_batch_through_few_slots_and_argument_checks(have_gpu, omc, orc):
    """Warm and cold nodes in one batch, more nodes than slots (a slot is re-initialised from the pool for its next node); every node's
    value equals its cold value; indices beyond the pool and a missing pool are argument errors."""
    A, mask = orc.make_instance(20, 24, 1, seed=3, kind="readme")
    eng = omc.Engine(A, mask, GAMMA, 1)
    P = omc.default_params(rho_scale=16.0, slots=3)
    with pytest.raises(omc.OmcError) as e:
        eng.stage([[]], "linear", P, load_from=[0])                        # no pool yet
    assert e.value.code == -3
    eng.state_pool_create(4)
    root = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P, save_to=[2])[0]
    kids = omc.pkg.bnb.make_children([], root, "linear", 1)
    lvl1 = eng.matrix_completion_SDP_relaxation(kids, "linear", params=P, load_from=[2, 2], save_to=[0, 1])
    gk = []; lf = []
    for i, (c, o) in enumerate(zip(kids, lvl1)):
        for c2 in omc.pkg.bnb.make_children(c, o, "linear", 1):
            gk.append(c2); lf.append(i if len(gk) % 3 else -1)            # every third grandchild starts cold
    warm = eng.matrix_completion_SDP_relaxation(gk, "linear", params=P, load_from=lf)
    cold = eng.matrix_completion_SDP_relaxation(gk, "linear", params=P)
    for w_, c_ in zip(warm, cold):
        assert_finite(w_)
        if w_["status_code"] == 0 and c_["status_code"] == 0:
            assert w_["objective"] == pytest.approx(c_["objective"], rel=OBJ_REL)
        assert w_["dual_bound"] <= c_["objective"] * (1 + 1e-5) + 1e-9 and c_["dual_bound"] <= w_["objective"] * (1 + 1e-5) + 1e-9   # both bounds valid for the same program
    with pytest.raises(omc.OmcError) as e:
        eng.stage(gk, "linear", P, load_from=[4] * len(gk))                # beyond the pool
    assert e.value.code == -3
    eng.close()


# ---- round 3: BASELINE config 5 (1000 x 1000, k = 2, 30 % observed): orders above 512 and problems that do not fit the LDS ----------
def test_altmin_global_slab_variant(have_gpu, omc, orc):
    """alternating_minimization on problems that do not fit the LDS (k_altmin / k_altmin_k on a per-problem global slab): bit-identical
    to the LDS-resident launch where both run, and equal to the oracle at the BASELINE config 5 size (1000 x 1000, k = 2: 144 KB of
    state per problem, over the 128 KB launch limit)."""
    for k, (n, m) in ((1, (30, 40)), (2, (16, 22))):
        A, mask = orc.make_instance(n, m, k, seed=90 + k, kind="lowrank", n_indices=int(0.5 * n * m))
        eng = omc.Engine(A, mask, GAMMA, k)
        U0 = orc.svd_rounding(np.where(mask, A, 0.0), k)
        x = np.linalg.qr(np.random.default_rng(5).standard_normal((n, 1)))[0][:, 0]
        dirs = orc.child_directions("linear", k)
        nodes = [[], [(x, U0 * 0.7, list(dirs[0]))]]
        a = eng.alternating_minimization([U0, U0], nodes, "linear", max_iters=30)
        eng.tuning_set("OMC_ALTMIN_NOLDS", "1")
        try:
            b = eng.alternating_minimization([U0, U0], nodes, "linear", max_iters=30)
        finally:
            eng.tuning_set("OMC_ALTMIN_NOLDS", None)
        for g, h in zip(a, b):
            assert g["n_iters"] == h["n_iters"] and np.array_equal(g["objectives"], h["objectives"])
            assert np.array_equal(g["U"], h["U"]) and np.array_equal(g["V"], h["V"])
        eng.close()
    A, mask, gamma, c = omc.pkg.data.config_instance(5, seed=0)
    n, m = A.shape
    inst = orc.Instance(A, mask, gamma, 2)
    eng = omc.Engine(A, mask, gamma, 2)
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 2)
    x = np.linalg.qr(np.random.default_rng(6).standard_normal((n, 1)))[0][:, 0]
    cuts = [(x, U0 * 0.8, list(orc.child_directions("linear", 2)[0]))]
    got = eng.alternating_minimization([U0, U0], [[], cuts], "linear", max_iters=4)
    for g, cl in zip(got, [[], cuts]):
        r = orc.alternating_minimization(inst, U0, cl, "linear", max_iters=4)
        assert g["n_iters"] == r["n_iters"] and g["converged"] == r["converged"]
        assert np.allclose(g["objectives"], r["objectives"], rtol=1e-8)
        assert np.allclose(g["U"] @ g["V"], r["U"] @ r["V"], atol=1e-5)
        assert g["master_objective"] == pytest.approx(orc.evaluate_objective(g["U"] @ g["V"], A, mask, gamma), rel=1e-10)
    eng.close()


def test_config5_node_evaluation(have_gpu, omc, orc):
    """One node of BASELINE config 5 (1000 x 1000, k = 2, 30 % observed, cone order 1002): the L2-resident eigen-kernel with a whole
    wave per rotation pair seeds the 16-vector block (Z slab in global memory above order 512), which then does the projections.
    100 iterations (a full solve is minutes): outputs finite, the Lagrangian bound valid against the best
    known feasible objective (the rounded altmin point), Y inside its cone up to the truncation."""
    A, mask, gamma, c = omc.pkg.data.config_instance(5, seed=0)
    n, m = A.shape
    eng = omc.Engine(A, mask, gamma, 2)
    P = omc.default_params(rho_scale=4.0, max_iters=100, check_every=25)
    o = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P, want_X=True)[0]
    assert_finite(o)
    assert o["iters"] == 100 and o["status_code"] in (1, 3) and o["feasible"]
    st = eng.subspace_stats()
    assert st["calls"] >= 20 and st["fallbacks"] <= 5, st
    U0 = orc.svd_rounding(np.where(mask, A, 0.0), 2)
    am = eng.alternating_minimization([U0], [[]], "linear", max_iters=20)[0]
    ub = am["master_objective"]
    assert o["dual_bound"] <= ub * (1 + 1e-9)                             # valid lower bound (any feasible rank-2 point is above it)
    assert o["X"].shape == (n, m) and o["Y"].shape == (n, n)
    w = np.linalg.eigvalsh(o["Y"])
    assert w[0] >= -1e-2 and w[-1] <= 1 + 1e-2 and np.trace(o["Y"]) <= 2 + 1e-6
    eng.close()


def _env_run(eng, omc, nodes, P, env):
    """One relaxation of `nodes` with the tuning knobs `env` set on the instance (omc_tuning_set: the library reads no environment after creation)."""
    for k_, v in env.items():
        eng.tuning_set(k_, v)
    try:
        return eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False)
    finally:
        for k_ in env:
            eng.tuning_set(k_, None)


def test_colprox_pair_kernel_against_one_column_kernel(have_gpu, omc, orc):
    """k_colprox_pair (two columns per wave, inverse by the symmetric sweep operator in registers) against k_colprox (one column per wave,
    L D L' in LDS) on the same nodes at a fixed iteration count: the two are different arithmetic for the same prox, so they agree to
    round-off amplified by the ADMM map, not bit for bit.  Second instance: odd m (unpaired last column), columns with more than 32
    observed rows (they stay with k_colprox: w.cp_solo) next to sparse ones, and an empty column."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0, max_iters=300, eps_gap=1e-14)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 3, "linear", params=omc.default_params(rho_scale=4.0))
    a = _env_run(eng, omc, nodes, P, {})
    b = _env_run(eng, omc, nodes, P, {"OMC_NO_COLPROX_PAIR": "1"})
    for x, y in zip(a, b):
        assert x["iters"] == y["iters"] == 300
        assert x["objective"] == pytest.approx(y["objective"], rel=1e-9) and x["dual_bound"] == pytest.approx(y["dual_bound"], rel=1e-8, abs=1e-8)
        assert np.allclose(x["Y"], y["Y"], atol=1e-9)
    eng.close()
    rng = np.random.default_rng(11)
    n, m = 40, 45
    dens = rng.choice([0.15, 0.5, 0.95], size=m)
    mask = rng.random((n, m)) < dens[None, :]
    mask[:, 7] = False                                   # an empty column
    mask[0, :7] = True; mask[1, 8:] = True               # every row observed somewhere
    U0 = rng.standard_normal((n, 1)); V0 = rng.standard_normal((1, m))
    A = (U0 @ V0 + 0.01 * rng.standard_normal((n, m))) * mask
    assert (mask.sum(0) > 32).any() and (mask.sum(0) <= 32).any()
    eng = omc.Engine(A, mask, GAMMA, 1)
    inst = orc.Instance(A, mask, GAMMA, 1)
    P = omc.default_params(rho_scale=4.0)
    got = eng.matrix_completion_SDP_relaxation([[]], "linear", params=P)[0]
    ref = orc.sdp_relaxation(inst, [], "linear", params=orc.RelaxParams(rho_scale=4.0))
    assert got["status_code"] == 0 and ref["termination_status"] == 0
    assert got["objective"] == pytest.approx(ref["objective"], rel=2e-6) and got["dual_bound"] == pytest.approx(ref["dual_bound"], rel=2e-6)
    eng.close()


def test_split_launch_of_the_full_eigen_kernel_is_bit_identical(have_gpu, omc):
    """The full eigen-kernel as a known-in-advance launch beside k_cone_sub plus an (almost empty) launch behind it, against one launch
    after k_cone_sub (OMC_NO_WS_SPLIT): the same kernel runs each slot either way, so every result is bit-identical; likewise event timing
    of every 8th iteration only (OMC_TIMING_STRIDE), which changes what is measured and not what is computed."""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0, max_iters=400)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 5, "linear", params=P)
    a = _env_run(eng, omc, nodes, P, {})
    b = _env_run(eng, omc, nodes, P, {"OMC_NO_WS_SPLIT": "1"})
    c_ = _env_run(eng, omc, nodes, P, {"OMC_TIMING_STRIDE": "8"})
    with pytest.raises(omc.OmcError):
        eng.tuning_set("OMC_NOT_A_KNOB", "1")                            # unknown knobs are refused, not ignored
    os.environ["OMC_NO_WS_SPLIT"] = "1"                                  # the environment is read at creation (and on request) only
    try:
        d_ = eng.matrix_completion_SDP_relaxation(nodes[:4], "linear", params=P, want_X=False)
    finally:
        del os.environ["OMC_NO_WS_SPLIT"]
    assert [o["iters"] for o in d_] == [o["iters"] for o in a[:4]]
    # hipGraph replay of the iteration body (batches of at most OMC_GRAPH_MAX nodes) computes what the eager launches compute
    e_ = _env_run(eng, omc, nodes[:4], P, {"OMC_GRAPH_MAX": "0"})
    for x, z in zip(a[:4], e_):
        assert (x["objective"], x["dual_bound"], x["iters"], x["status_code"]) == (z["objective"], z["dual_bound"], z["iters"], z["status_code"])
        assert np.array_equal(x["Y"], z["Y"])
    for x, y, z in zip(a, b, c_):
        for key in ("objective", "dual_bound", "iters", "status_code"):
            assert x[key] == y[key] == z[key], key
        assert np.array_equal(x["Y"], y["Y"]) and np.array_equal(x["Y"], z["Y"])
    eng.close()


def test_config4_root_certified(have_gpu, omc, orc):
    """BASELINE config 4 (500 x 500, rank 2) root run to certification: the two-sided 1e-6 gap, and the returned (X, Y, Theta, U) checked
    against every cone and row of the reference's program (OMC.jl:1554-1685) at the tolerance the small cases use -- the oracle's
    primal_residuals on the GPU's point (the oracle's own solve would take hours at this size; the residual evaluation takes seconds)."""
    A, mask, gamma, c = omc.pkg.data.config_instance(4, seed=0)
    n, m = A.shape
    eng = omc.Engine(A, mask, gamma, c["k"])
    P = omc.default_params(rho_scale=4.0, max_iters=1500, breakpoints=2)
    g = eng.matrix_completion_SDP_relaxation([[]], "linear3", params=P, want_Theta=True)[0]
    assert_finite(g)
    assert g["status_code"] == 0, (g["termination_status"], g["iters"], g["objective"], g["dual_bound"])
    assert abs(g["objective"] - g["dual_bound"]) <= 1.01e-6 * max(1.0, abs(g["objective"]))
    inst = orc.Instance(A, mask, gamma, c["k"])
    rows = orc.build_rows(inst, [], "linear3")
    Theta = 0.5 * (g["Theta"] + g["Theta"].T)
    res = orc.primal_residuals(inst, rows, g["Y"], g["U"], g["X"], Theta)
    assert res["max"] <= 2e-5 * max(1.0, np.abs(Theta).max()), res
    assert orc.compute_SDP_relaxation_objective(g["X"], Theta, A, mask, gamma) == pytest.approx(g["objective"], rel=1e-8)
    st = eng.subspace_stats()
    assert st["fallbacks"] <= 10 and st["calls"] >= g["iters"] // 2, st
    eng.close()


def test_early_slow_progress_prediction_on_a_frontier_sample(have_gpu, omc):
    """The early SLOW_PROGRESS rule (k_check_final: the gap, at the geometric rate of the last checks, cannot close before max_iters):
    the nodes of a cold config-2 depth-8 frontier that it returns early are relaxed again with the rule off.  The rule is a prediction: measured
    here, 5 of the 17 early-stopped nodes (of 256) would have been certified before max_iters -- the test bounds that share by one half and
    prints it -- and what the early return reported must have been a valid bound: below the objective of the long run's (better converged)
    point.  (With warm starts, the bench default, the rule never fires: the same 51 of 2048 nodes stay uncertified with it on or off.)"""
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 8, "linear", params=P)
    out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False, want_Y=False)
    early = [i for i, o in enumerate(out) if o["status_code"] == 1 and o["iters"] < P.max_iters]
    assert len(early) >= 3, "the frontier no longer holds early-stopped nodes: deepen it"
    sample = early[:24]
    P0 = omc.default_params(rho_scale=4.0, early_stop_factor=0.0)
    late = eng.matrix_completion_SDP_relaxation([nodes[i] for i in sample], "linear", params=P0, want_X=False, want_Y=False)
    wrong = 0
    for i, o in zip(sample, late):
        assert o["iters"] >= out[i]["iters"]
        assert out[i]["dual_bound"] <= o["objective"] * (1 + 1e-6) + 1e-9           # the early bound was valid
        if o["status_code"] == 0:
            wrong += 1
        else:
            assert o["status_code"] == 1
    print(f"early SLOW_PROGRESS: {len(early)} of {len(nodes)} nodes; of {len(sample)} re-run without the rule {wrong} were certified before max_iters")
    assert wrong <= len(sample) // 2
    eng.close()


def test_left_singular_vectors_on_the_device(have_gpu, omc, orc):
    """omc_left_singular_batch = svd(X).U[:, 1:k] (OMC.jl:524, 564, 921): Gram product on the matrix cores + the top-k eigen-kernel, against
    numpy's SVD for the matrices the driver rounds -- a rank-k product U V, the zero-filled A of the root, a relaxation's X -- at k = 1 and 2
    (compared as projectors U U', and entry-wise after the canonical sign where the singular values are distinct)."""
    for k in (1, 2):
        A, mask = orc.make_instance(40, 52, k, seed=30 + k, kind="lowrank", n_indices=int(0.4 * 40 * 52))
        eng = omc.Engine(A, mask, GAMMA, k)
        rng = np.random.default_rng(k)
        U = rng.standard_normal((40, k)); V = rng.standard_normal((k, 52))
        root = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc.default_params(rho_scale=4.0))[0]
        mats = [U @ V, np.where(mask, A, 0.0), root["X"]]
        got = eng.left_singular(mats)
        for X, g in zip(mats, got):
            Uf, sv, _ = np.linalg.svd(X, full_matrices=False)
            ref = Uf[:, :k]
            assert np.allclose(g.T @ g, np.eye(k), atol=1e-10)
            assert np.allclose(g @ g.T, ref @ ref.T, atol=1e-8 * max(1.0, sv[0] / max(sv[k - 1] - (sv[k] if len(sv) > k else 0.0), 1e-12)))
            for j in range(k):
                r = ref[:, j] * np.sign(ref[np.argmax(np.abs(ref[:, j])), j])
                if k == 1 or abs(sv[0] - sv[1]) > 1e-6 * sv[0]:
                    assert np.allclose(g[:, j], r, atol=1e-7)
        # the driver's rounding goes through it
        Xk, Uk = omc.pkg.bnb.rank_k_projection(root["X"], k, eng)
        assert np.linalg.matrix_rank(Xk, tol=1e-9) <= k and np.allclose(Uk, got[2])
        eng.close()


def test_append_nodes_to_a_staged_and_to_a_running_batch(have_gpu, omc):
    """omc_relax_reserve / omc_relax_append: nodes added to a staged batch before the solve, and nodes added while the submitted solve is
    running (the loop hands them to free slots at its next check), come back with the results the same nodes have when they are all staged
    together -- bit for bit, a node's relaxation does not depend on the slot or the moment it starts.  With warm starts from the pool as well;
    refused beyond the reserved capacity, with more cuts than reserved, and after the solve has ended."""
    import time
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0, slots=32)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 6, "linear", params=omc.default_params(rho_scale=4.0))
    assert len(nodes) == 64
    ref = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False, want_Y=False)
    key = lambda o: (o["objective"], o["dual_bound"], o["iters"], o["status_code"])
    # (a) appended before the solve starts
    eng.reserve(48, 6)
    eng.stage(nodes[:16], "linear", P)
    eng.append(nodes[16:40], "linear")
    eng.append(nodes[40:], "linear")
    eng.solve()
    got = eng.fetch(want_Y=False, want_X=False)
    assert [key(o) for o in got] == [key(o) for o in ref]
    # (b) appended while the solve is running (16 nodes through 32 slots: slots are idle until the first append arrives)
    eng.reserve(48, 6)
    eng.stage(nodes[:16], "linear", P)
    eng.submit()
    eng.append(nodes[16:40], "linear")
    time.sleep(0.05)
    eng.append(nodes[40:], "linear")
    assert eng.poll()["nodes_total"] == 64
    eng.wait()
    got = eng.fetch(want_Y=False, want_X=False)
    assert [key(o) for o in got] == [key(o) for o in ref]
    with pytest.raises(omc.OmcError):
        eng.append(nodes[:1], "linear")                                   # the solve has ended
    # (c) limits
    eng.reserve(4, 2)
    eng.stage([[]] * 4, "linear", P)
    with pytest.raises(omc.OmcError):
        eng.append(nodes[:1], "linear")                                   # six cuts, two reserved
    with pytest.raises(omc.OmcError):
        eng.append([[]] * 5, "linear")                                    # beyond the capacity
    eng.append([nodes[0][:2]] * 4, "linear")
    eng.solve()
    out = eng.fetch(want_Y=False, want_X=False)
    assert len(out) == 8 and all(key(o) == key(out[0]) for o in out[1:4]) and all(key(o) == key(out[4]) for o in out[5:])
    # (d) warm starts from the pool for appended nodes
    parents = [list(cuts[:-1]) for cuts in nodes[::2]]
    eng.state_pool_create(len(parents))
    eng.matrix_completion_SDP_relaxation(parents, "linear", params=P, want_X=False, want_Y=False, save_to=list(range(len(parents))))
    lf = [i // 2 for i in range(len(nodes))]
    refw = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False, want_Y=False, load_from=lf)
    eng.reserve(48, 6)
    eng.stage(nodes[:16], "linear", P, load_from=lf[:16])
    eng.submit()
    eng.append(nodes[16:], "linear", load_from=lf[16:])
    eng.wait()
    gotw = eng.fetch(want_Y=False, want_X=False)
    assert [key(o) for o in gotw] == [key(o) for o in refw]
    assert sum(o["iters"] for o in gotw) < sum(o["iters"] for o in ref)   # and they did start warm
    eng.close()


def test_queue_driven_loop_append_and_fetch_done(have_gpu, omc):
    """omc_relax_fetch_done + omc_relax_append: a host loop in the shape of the reference's (OMC.jl:700-719) -- take the results of the nodes that
    have finished so far, push new nodes for them -- against one staged batch: every node comes back exactly once through fetch_done, in
    finishing order, with the values the batch call returns, while the solve keeps running until the host stops feeding it."""
    import time
    A, mask, gamma, c = omc.pkg.data.config_instance(2, seed=0)
    eng = omc.Engine(A, mask, gamma, 1)
    P = omc.default_params(rho_scale=4.0, slots=16)
    nodes, _ = omc.pkg.bnb.expand_frontier(eng, 6, "linear", params=omc.default_params(rho_scale=4.0))
    ref = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False, want_Y=False)
    eng.reserve(len(nodes) - 8, 6)
    eng.stage(nodes[:8], "linear", P)
    eng.hold(True)                                                        # the solve waits for pushes when it runs dry
    eng.submit()
    sent, seen, deadline = 8, {}, time.time() + 120
    while len(seen) < len(nodes) and time.time() < deadline:
        got = eng.fetch_done()
        for o in got:
            assert o["node"] not in seen
            seen[o["node"]] = o
        if got and sent < len(nodes):                                     # two new nodes per finished one, as a split would push
            more = min(2 * len(got), len(nodes) - sent)
            eng.append(nodes[sent:sent + more], "linear")
            sent += more
        if not got:
            time.sleep(0.002)
    assert eng.poll()["running"]                                          # every node is back and the solve is still open
    eng.hold(False)
    eng.wait()
    assert sorted(seen) == list(range(len(nodes)))
    for i, o in seen.items():
        r = ref[i]
        assert (o["objective"], o["dual_bound"], o["iters"], o["status_code"]) == (r["objective"], r["dual_bound"], r["iters"], r["status_code"])
        assert np.array_equal(o["U"], r["U"]) and np.array_equal(o["breakpoint_vec"], r["breakpoint_vec"])
    assert eng.fetch_done() == []
    eng.close()


def test_streaming_branch_and_bound_against_the_round_based_driver(have_gpu, omc, orc):
    """bnb_stream.branch_and_bound_streaming (one running solve fed through omc_relax_append / omc_relax_fetch_done / omc_relax_hold) against
    bnb.branch_and_bound on an instance whose tree branches, the same time budget each: the two runs bracket the same optimum (each lower
    bound below the other's incumbent), and the streaming run keeps the invariants of SURVEY 8c (LB monotone, LB <= UB, incumbent =
    evaluate_objective of a rank-k X, counter identity) and warm-starts its nodes."""
    A, mask = orc.make_instance(14, 18, 1, seed=5, kind="lowrank", n_indices=int(0.35 * 14 * 18), noise=0.15)
    eng = omc.Engine(A, mask, GAMMA, 1)
    a, ia = omc.pkg.bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=8.0, batch=64, rho_scale=8.0)
    b, ib = omc.pkg.bnb_stream.branch_and_bound_streaming(eng, A, mask, gap=1e-4, time_limit=8.0, slots=64, rho_scale=8.0)
    assert a["lower_bound"] <= b["objective"] * (1 + 1e-9) and b["lower_bound"] <= a["objective"] * (1 + 1e-9)
    assert b["objective"] <= a["objective"] * (1 + 0.05)                      # the same altmin schedule finds a comparable incumbent
    c = ib["run_details"]; log = np.array(ib["run_log"])
    assert c["nodes_relax_infeasible"] + c["nodes_relax_feasible"] == c["nodes_explored"] and c["nodes_explored"] > 50
    assert (np.diff(log[:, 3]) >= -1e-9).all() and b["lower_bound"] <= b["objective"] * (1 + 1e-9)
    assert np.linalg.matrix_rank(b["X"], tol=1e-8) <= 1
    assert b["objective"] == pytest.approx(orc.evaluate_objective(b["X"], A, mask, GAMMA), rel=1e-10)
    assert c["warm_started"] > 0.5 * c["nodes_explored"]
    print("round-based:", ia["run_details"]["nodes_explored"], "nodes, lb", a["lower_bound"], "ub", a["objective"], "; streaming:", c["nodes_explored"], "nodes, lb", b["lower_bound"], "ub", b["objective"], "epochs", c["epochs"], "warm", c["warm_started"])
    eng.close()
