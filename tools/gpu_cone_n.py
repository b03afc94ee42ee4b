"""Root relaxation of a lowrank instance of order n (rank k): timing per kernel class, subspace statistics (large-order path probe)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd, omc_oracle as orc
n = int(os.environ.get("N", "600")); k = int(os.environ.get("K", "2")); IT = int(os.environ.get("ITERS", "50")); frac = float(os.environ.get("FRAC", "0.2"))
os.environ["OMC_GRAPH_MAX"] = "0"
A, mask = orc.make_instance(n, n, k, n_indices=int(frac * n * n), seed=0, noise=0.01)
t0 = time.time(); eng = omc_amd.Engine(A, mask, 80.0, k); t_create = time.time() - t0
P = omc_amd.default_params(rho_scale=4.0, max_iters=IT, check_every=min(25, IT), breakpoints=2 if k > 1 else 1)
t0 = time.time(); eng.stage([[]], "linear", P); t_stage = time.time() - t0
t0 = time.time(); eng.solve(); t_solve = time.time() - t0
r = eng.fetch(want_Y=True, want_X=False)[0]
print(json.dumps(dict(n=n, k=k, iters=r["iters"], create_s=round(t_create, 2), stage_s=round(t_stage, 2), solve_s=round(t_solve, 2), objective=r["objective"], dual_bound=r["dual_bound"], status=r["status_code"],
                      finite=bool(np.isfinite(r["Y"]).all()), kernels={k_: (round(v["ms"], 1), v["launches"]) for k_, v in eng.kernel_stats().items() if v["launches"]}, sub=eng.subspace_stats())), flush=True)
