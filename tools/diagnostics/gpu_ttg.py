"""Wall-clock of the whole B&B (root altmin, penalty autotune, tree) to gap <= 1e-4 on config 2, seeds 0-2 (as bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
for rep in range(2):
    for sd in (0, 1, 2):
        A, mask, g, c = omc_amd.pkg.data.config_instance(2, seed=sd)
        e = omc_amd.Engine(A, mask, g, c["k"])
        t1 = time.perf_counter()
        sol, inst = omc_amd.pkg.bnb.branch_and_bound(e, A, mask, gap=1e-4, time_limit=120.0, batch=128, disjunctive_cuts_type=c["cut_type"])
        print("rep", rep, "seed", sd, "%.3fs" % (time.perf_counter() - t1), "gap %.1e" % sol["gap"], "nodes relaxed", inst["run_details"]["nodes_relax_feasible"], "rho_scale", inst["run_details"]["rho_scale"], flush=True)
        e.close()
