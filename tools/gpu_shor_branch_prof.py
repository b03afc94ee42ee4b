"""Shor-mode root of the branching instance (100 x 100, 2482 four-entry minors): kernel classes of one solve (eager launches, HIP events)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
data = omc_amd.pkg.data
A, mask = data.branching_instance(seed=0)
eng = omc_amd.Engine(A, mask, 80.0, 1)
B = int(os.environ.get("B", "1"))
mi = eng.generate_rank1_matrix_completion_Shor_constraints_indexes([4])
for gm in (os.environ.get("GRAPH", "0"),):
    eng.tuning_set("OMC_GRAPH_MAX", gm)
    P = omc_amd.default_params(rho_scale=1.0, eps_gap=1e-5, max_iters=6000, slots=B)
    eng.stage_shor([[]] * B, [(mi, None)] * B, "linear", P)
    t0 = time.time(); eng.solve(); t = time.time() - t0
    o = eng.fetch(want_Y=False, want_X=False)
    ks = {c: (round(v["ms"], 1), v["launches"]) for c, v in eng.kernel_stats().items() if v["launches"]}
    print(json.dumps(dict(graph_max=gm, B=B, seconds=round(t, 2), iters=o[0]["iters"], status=o[0]["status_code"], obj=o[0]["objective"], lb=o[0]["dual_bound"], ms_per_iter=round(t / o[0]["iters"] * 1e3, 3),
                          kernels=ks, shor_sub=eng.shor_subspace_stats(), sub=eng.subspace_stats())), flush=True)
