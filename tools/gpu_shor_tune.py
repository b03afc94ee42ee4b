"""Penalty scan of the Shor-mode splitting on the GPU (development tool): iterations to certify 1e-5 / final gap per (rho, r4, r5, relax)."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd
import omc_oracle as orc, omc_oracle_shor as sh

insts = []
for (n, m, nidx, seed, noise) in [(20, 24, 150, 1, 0.05), (20, 24, 150, 2, 0.1), (20, 24, 150, 3, 0.1), (20, 24, 150, 5, 0.1), (30, 36, 320, 1, 0.1), (30, 36, 320, 2, 0.05)]:
    A, mask = orc.make_instance(n, m, 1, n_indices=nidx, seed=seed, noise=noise)
    minors, _ = sh.driver_shor_lists(mask, (4,))
    insts.append((A, mask, minors))
grid = [(0.05, 20, 2, 1.6), (0.05, 20, 2, 1.8), (0.05, 20, 2, 1.0), (0.02, 20, 2, 1.6), (0.1, 20, 2, 1.6), (0.05, 5, 2, 1.6), (0.05, 60, 2, 1.6), (0.05, 20, 0.5, 1.6), (0.05, 20, 8, 1.6),
        (0.02, 50, 5, 1.6), (0.1, 10, 1, 1.6), (0.2, 5, 1, 1.6), (0.05, 20, 2, 1.9)]
eps = float(os.environ.get("EPS", "1e-5")); mx = int(os.environ.get("MAXIT", "8000")); bump = int(os.environ.get("BUMP", "1"))
for (rho, r4, r5, rx) in grid:
    row = []
    for (A, mask, minors) in insts:
        eng = omc_amd.Engine(A, mask, 80.0, 1)
        p = omc_amd.default_params(eps_gap=eps, max_iters=mx, relax=rx, bump_max=bump)
        r = eng.matrix_completion_SDP_relaxation([[]], "linear", p, add_Shor_valid_inequalities=True, shor_info=[(minors, None)], shor_penalties=(rho, r4, r5), want_Y=False, want_X=False)[0]
        row.append(f"{r['iters']}{'*' if r['status_code'] == 0 else ''}/{(r['objective'] - r['dual_bound']) / abs(r['objective']):.0e}")
        eng.close()
    print((rho, r4, r5, rx), " ".join(row), flush=True)
