"""Golden fixtures (tests/golden/*.npz, produced by tools/make_golden.py with the oracle; the reference cannot run
here).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
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
            assert o["objective"] - o["dual_bound"] <= 1.01e-6 * max(1.0, abs(o["objective"]))
        assert o["lambda_min"][0] == pytest.approx(float(z["lmin"][b]), abs=1e-5)
        assert eng.evaluate_objective(o["X"]) == pytest.approx(float(z["eval_obj"][b]), rel=1e-5)
    eng.close()
