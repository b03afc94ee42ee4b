"""The compiled single-thread restatement of the node relaxation (oracle/omc_cpu_ref.cpp: bench.py's cpu_baseline) against the numpy oracle
it restates: same iteration counts, objectives, bounds and Y on the root and on cut nodes.  CPU only (test infrastructure vs test
infrastructure: the product path is not involved)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import omc_oracle as orc      # noqa: E402
import omc_cpu_ref as cref    # noqa: E402

GAMMA = 80.0


@pytest.mark.parametrize("n,m,k,seed", [(12, 16, 1, 3), (10, 14, 2, 5)])
def test_cpu_ref_matches_numpy_oracle(n, m, k, seed):
    A, mask = orc.make_instance(n, m, k, seed=seed, kind="lowrank", n_indices=int(0.5 * n * m))
    inst = orc.Instance(A, mask, GAMMA, k)
    P = orc.RelaxParams(); P.rho_scale = 4.0
    root = orc.sdp_relaxation(inst, [], "linear", params=P)
    x, _ = orc.breakpoint_vector(root["Y"], root["U"])
    kids = [[(x, root["U"], list(d))] for d in orc.child_directions("linear", k)]
    nodes = [[]] + kids
    got, Y0 = cref.relax_nodes(inst, nodes, "linear", params=P, threads=1)
    for g, cuts in zip(got, nodes):
        r = orc.sdp_relaxation(inst, cuts, "linear", params=P)
        assert g["status_code"] == r["termination_status"]
        assert abs(g["iters"] - r["iters"]) <= P.check_every            # the same algorithm; a check boundary may differ by round-off
        if r["termination_status"] != orc.OMC_INFEASIBLE:
            assert g["objective"] == pytest.approx(r["objective"], rel=2e-6)
            assert g["dual_bound"] == pytest.approx(r["dual_bound"], rel=2e-6)
            assert g["dual_bound"] <= g["objective"] * (1 + 1e-6) + 1e-9
    assert np.allclose(Y0, root["Y"], atol=1e-5)
    # OpenMP over the nodes returns what the serial loop returns
    if cref.load().omc_cpu_ref_openmp() > 0:
        got2, _ = cref.relax_nodes(inst, nodes, "linear", params=P, threads=2)
        for a, b in zip(got, got2):
            assert a == b
