"""Which 100x100 instances branch?  Runs the driver counterpart for a few (noise, observed fraction) pairs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
for noise, frac in ((0.15, 0.2), (0.2, 0.2), (0.1, 0.1), (0.15, 0.1), (0.1, 0.06)):
    A, mask = data.branching_instance(seed=0, noise=noise, frac=frac)
    eng = omc_amd.Engine(A, mask, 80.0, 1)
    t0 = time.perf_counter()
    sol, inst = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=30.0, batch=256)
    rd = inst["run_details"]
    print("noise %.2f frac %.2f: %.1fs gap %.2e lb %.4f ub %.4f relaxed %d explored %d total %d relax_s %.1f altmin_s %.1f" % (noise, frac, time.perf_counter() - t0, sol["gap"], sol["lower_bound"], sol["objective"],
          rd["nodes_relax_feasible"], rd["nodes_explored"], rd["nodes_total"], rd["solve_time_relaxation"], rd["solve_time_altmin"]), flush=True)
    eng.close()
