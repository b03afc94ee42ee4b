import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
cfg = int(os.environ.get("CFG", 3))
A, mask, gamma, c = omc_amd.pkg.data.config_instance(cfg, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
for sc in [2.0, 8.0]:
    t = time.time()
    out = eng.matrix_completion_SDP_relaxation([[]], c["cut_type"], params=omc_amd.default_params(rho_scale=sc, max_iters=int(os.environ.get("ITERS", 400))), want_X=False)[0]
    print("cfg", cfg, "n", c["n"], "scale", sc, out["termination_status"], out["iters"], "obj %.7f lb %.7f" % (out["objective"], out["dual_bound"]), "%.2fs" % (time.time() - t), eng.kernel_stats(), eng.solver_info(), flush=True)
