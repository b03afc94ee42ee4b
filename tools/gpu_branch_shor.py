"""Does the Shor-mode relaxation close the tree of the branching instance (100 x 100 rank 1, noise 0.3, 10 % observed) faster than the
disjunctive cuts alone?  bnb.branch_and_bound with and without add_Shor_valid_inequalities, same time limit."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
TL = float(os.environ.get("TL", "30"))
OUT = open(os.path.join(ROOT, "gpurun_out", "branch_shor.txt"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=OUT, flush=True)
if os.environ.get("INSTANCE", "branching") == "config1":
    A, mask, _g, _c = data.config_instance(1, seed=0)
else:
    A, mask = data.branching_instance(seed=0)
eng = omc_amd.Engine(A, mask, 80.0, 1)
ONLY = os.environ.get("ONLY")
for cl in ((4,), (3, 4)):
    idx = eng.generate_rank1_matrix_completion_Shor_constraints_indexes(list(cl))
    say("classes", cl, "minors", len(idx))
runs = [("base", dict()),
        ("shor[4] static", dict(add_Shor_valid_inequalities=True, Shor_valid_inequalities_noisy_rank1_num_entries_present=(4,))),
        ("shor iterative", dict(add_Shor_valid_inequalities=True, add_Shor_valid_inequalities_iterative=True, Shor_valid_inequalities_noisy_rank1_num_entries_present=(3, 4)))]
for name, kw in runs:
    if ONLY and ONLY not in name:
        continue
    t0 = time.time()
    try:
        if kw:
            kw = dict(kw, shor_params=omc_amd.default_params(rho_scale=1.0, eps_gap=1e-5, max_iters=int(os.environ.get("SHOR_ITERS", "3000")), time_limit=15.0))
        say("start", name)
        sol, inst = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=TL, batch=int(os.environ.get("BATCH", "32")), **kw)
        rd = inst["run_details"]
        say(name, json.dumps(dict(seconds=round(time.time() - t0, 1), gap=sol["gap"], lower=sol["lower_bound"], upper=sol["objective"], nodes=rd["nodes_relax_feasible"],
                                    relax_s=round(rd["solve_time_relaxation"], 1)), default=float))
    except Exception as ex:
        say(name, "failed:", repr(ex))
