import os, sys, time, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
bnb, bs, data = omc_amd.pkg.bnb, omc_amd.pkg.bnb_stream, omc_amd.pkg.data
if os.environ.get("MAIN", "0") == "1":      # as bench.py: a config-2 engine that has streamed a frontier through 1024 slots lives beside the branching one
    A2, m2, g2, c2 = data.config_instance(2, seed=0)
    e0 = omc_amd.Engine(A2, m2, g2, 1)
    P0 = omc_amd.default_params(rho_scale=4.0, slots=1024)
    nd0, _ = bnb.expand_frontier(e0, 9, "linear", params=P0)
    e0.stage(nd0 * 2, "linear", P0); e0.solve(); e0.fetch(want_Y=False, want_X=False)
    print("main engine done", flush=True)
A, mask = data.branching_instance(seed=0)
eng = omc_amd.Engine(A, mask, 80.0, 1)
mode = os.environ.get("MODE", "stream")
if mode == "both":
    s, i = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=float(os.environ.get("TL", "5")), batch=256)
    print("round done", i["run_details"]["nodes_explored"], flush=True)
    s, i = bs.branch_and_bound_streaming(eng, A, mask, gap=1e-4, time_limit=float(os.environ.get("TL", "5")), slots=1024)
    print("stream done", i["run_details"]["nodes_explored"], i["run_details"]["epochs"], flush=True)
    sols, insts = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=60.0, batch=32, add_Shor_valid_inequalities=True, Shor_valid_inequalities_noisy_rank1_num_entries_present=(4,),
                                       shor_params=omc_amd.default_params(rho_scale=1.0, eps_gap=1e-5, max_iters=6000, time_limit=30.0))
    print("shor bnb done", sols["gap"], flush=True)
    sys.exit(0)
elif mode == "stream":
    s, i = bs.branch_and_bound_streaming(eng, A, mask, gap=1e-4, time_limit=float(os.environ.get("TL", "5")), slots=1024)
    print("stream done", i["run_details"]["nodes_explored"], flush=True)
elif mode == "round":
    s, i = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=5.0, batch=256)
    print("round done", flush=True)
mi = eng.generate_rank1_matrix_completion_Shor_constraints_indexes([4])
print("minors", len(mi), flush=True)
P = omc_amd.default_params(rho_scale=1.0, eps_gap=1e-5, max_iters=300)
eng.stage_shor([[]], [(mi, None)], "linear", P)
print("staged", flush=True)
eng.solve()
print("solved", eng.fetch(want_Y=False, want_X=False)[0]["objective"], flush=True)
