"""BASELINE config 5 shape (1000 x 1000, rank 2, 30 % observed) in the disjunctive form: a capped root relaxation (timing probe)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
data = omc_amd.pkg.data
os.environ["OMC_GRAPH_MAX"] = "0"
A, mask, gamma, c = data.config_instance(5, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
IT = int(os.environ.get("ITERS", "2"))
P = omc_amd.default_params(rho_scale=4.0, max_iters=IT, breakpoints=2, check_every=int(os.environ.get("CHECK", str(IT))))
t0 = time.time()
eng.stage([[]], c["cut_type"], P)
t1 = time.time(); eng.solve(); t = time.time() - t1
print(json.dumps(dict(n=c["n"], k=c["k"], iters=IT, stage_s=round(t1 - t0, 2), solve_s=round(t, 2), kernels={k_: (round(v["ms"], 1), v["launches"]) for k_, v in eng.kernel_stats().items() if v["launches"]},
                      sub=eng.subspace_stats(), info={k_: float(v) for k_, v in eng.solver_info().items()})), flush=True)
