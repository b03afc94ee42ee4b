"""Golden fixtures (tests/golden/*.npz, produced by tools/make_golden.py with the oracle; the reference cannot run
here).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(f).startswith(("shor_", "altmin_")))
SHOR = os.path.join(HERE, "golden", "shor_9x13_k2.npz")
DIR_NAMES = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
REL = 2e-6   # objective tolerance: both sides are certified to 1e-6 of the optimum when OPTIMAL


def load(f):
    z = np.load(f, allow_pickle=False)
    cuts = [(z["cut_x"][l], z["cut_U"][l], [DIR_NAMES[int(c)] for c in z["cut_dir"][l]]) for l in range(len(z["cut_x"]))]
    nodes = [cuts[: int(L)] for L in z["node_L"]]
    return z, nodes


def test_fixtures_exist():
    assert len(FILES) >= 3


@pytest.mark.parametrize("f", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(f, orc):
    z, nodes = load(f)
    inst = orc.Instance(z["A"], z["mask"], float(z["gamma"]), int(z["k"]))
    for b in (0, len(nodes) - 2):       # two nodes per fixture keep the CPU suite short
        if int(z["iters"][b]) > 1000:
            continue
        r = orc.sdp_relaxation(inst, nodes[b], str(z["cut_type"]), params=orc.RelaxParams(rho_scale=float(z["rho_scale"])), want_certificate=False)
        assert r["objective"] == pytest.approx(float(z["objective"][b]), rel=1e-9)
        assert r["dual_bound"] == pytest.approx(float(z["dual_bound"][b]), rel=1e-9)
        assert r["iters"] == int(z["iters"][b]) and r["termination_status"] == int(z["status"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("f", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(f, omc):
    z, nodes = load(f)
    eng = omc.Engine(z["A"], z["mask"], float(z["gamma"]), int(z["k"]))
    out = eng.matrix_completion_SDP_relaxation(nodes, str(z["cut_type"]), params=omc.default_params(rho_scale=float(z["rho_scale"])))
    for b, o in enumerate(out):
        assert o["objective"] == pytest.approx(float(z["objective"][b]), rel=REL), (b, o["termination_status"])
        if int(z["status"][b]) == 0:
            assert o["status_code"] == 0
            assert o["dual_bound"] == pytest.approx(float(z["dual_bound"][b]), rel=REL)
            assert abs(o["objective"] - o["dual_bound"]) <= 1.01e-6 * max(1.0, abs(o["objective"]))
        assert o["lambda_min"][0] == pytest.approx(float(z["lmin"][b]), abs=1e-5)
        assert eng.evaluate_objective(o["X"]) == pytest.approx(float(z["eval_obj"][b]), rel=1e-5)
    eng.close()


def test_oracle_reproduces_shor_golden(orc):
    z = np.load(SHOR, allow_pickle=False)
    for p in (4, 3, 2, 1, 0):
        assert np.array_equal(np.array(orc.shor_constraints_indexes(z["mask"], [p]), dtype=np.int64).reshape(-1, 4), z["idx_%d" % p])
    assert np.array_equal(np.array(orc.shor_constraints_indexes(z["mask"], [2, 4]), dtype=np.int64).reshape(-1, 4), z["idx_2_4"])
    ex = [tuple(int(v) for v in t) for t in z["existing"]]
    for nm, e in (("first", []), ("second", ex)):
        got = orc.violated_shor_minors(z["X3"], z["mask"], [4, 3], e, 12)
        assert np.array_equal(np.array([t for _, t in got], dtype=np.int64), z[nm + "_minors"])
        assert np.array_equal(np.array([s for s, _ in got]), z[nm + "_scores"])


@pytest.mark.gpu
def test_hip_reproduces_shor_golden(omc):
    z = np.load(SHOR, allow_pickle=False)
    eng = omc.Engine(z["A"], z["mask"], 80.0, int(z["X3"].shape[0]))
    for p in (4, 3, 2, 1, 0):
        assert np.array_equal(eng.generate_rank1_matrix_completion_Shor_constraints_indexes([p]), z["idx_%d" % p])
    assert np.array_equal(eng.generate_rank1_matrix_completion_Shor_constraints_indexes([2, 4]), z["idx_2_4"])
    ex = [tuple(int(v) for v in t) for t in z["existing"]]
    for nm, e in (("first", []), ("second", ex)):
        got = eng.generate_violated_Shor_minors(z["X3"], [4, 3], e, 12)
        assert np.array_equal(np.array([t for _, t in got], dtype=np.int64), z[nm + "_minors"])
        assert np.array_equal(np.array([s for s, _ in got]), z[nm + "_scores"])
    eng.close()
