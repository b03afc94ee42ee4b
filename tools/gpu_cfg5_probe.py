"""BASELINE config 5 shape (1000 x 1000, rank 2, 30 % observed) in the disjunctive form: a capped root relaxation (timing probe)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
data = omc_amd.pkg.data
A, mask, gamma, c = data.config_instance(5, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
IT = int(os.environ.get("ITERS", "25"))
P = omc_amd.default_params(rho_scale=4.0, max_iters=IT, breakpoints=2, check_every=25)
t0 = time.time()
r = eng.matrix_completion_SDP_relaxation([[]], c["cut_type"], params=P, want_X=False)[0]
t = time.time() - t0
print(json.dumps(dict(n=c["n"], k=c["k"], iters=r["iters"], seconds=round(t, 2), objective=r["objective"], dual_bound=r["dual_bound"], status=r["status_code"],
                      finite=bool(np.isfinite(r["Y"]).all()), sub=eng.subspace_stats(), info={k_: float(v) for k_, v in eng.solver_info().items()})))
