"""The driver counterpart on the README instance (config 1: 50 x 50 noise, B&B really branches): nodes per second and where time goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask, g, c = omc_amd.pkg.data.config_instance(1, seed=0)
e = omc_amd.Engine(A, mask, g, c["k"])
rs, _ = omc_amd.pkg.bnb.autotune_rho_scale(e, c["cut_type"])
for batch, acc in ((512, 0), (512, 1)):
    t1 = time.perf_counter()
    sol, inst = omc_amd.pkg.bnb.branch_and_bound(e, A, mask, gap=1e-4, time_limit=float(sys.argv[1]) if len(sys.argv) > 1 else 20.0, batch=batch, disjunctive_cuts_type=c["cut_type"],
                                                 rho_scale=rs, params=omc_amd.default_params(rho_scale=rs, accel=acc))
    el = time.perf_counter() - t1; d = inst["run_details"]
    print("accel", acc, "batch %d: %.1fs, nodes relaxed %d (%.0f/s), explored %d, total %d; gap %.3f; LB %.4f UB %.4f; relaxation %.1fs altmin %.1fs host %.1fs" % (
        batch, el, d["nodes_relax_feasible"], d["nodes_relax_feasible"] / el, d["nodes_explored"], d["nodes_total"], sol["gap"], sol["lower_bound"], sol["objective"],
        d["solve_time_relaxation"], d["solve_time_altmin"], el - d["solve_time_relaxation"] - d["solve_time_altmin"]), flush=True)
