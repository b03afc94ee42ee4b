"""Total ADMM iterations of one frontier as a function of the over-relaxation and of the ratio of the column penalty (config 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P0 = omc_amd.default_params(rho_scale=4.0, slots=1024)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 9, c["cut_type"], params=P0)
for relax, rf, rs in ((1.6, 0.1, 4.0), (1.7, 0.1, 4.0), (1.8, 0.1, 4.0), (1.9, 0.1, 4.0), (1.6, 0.2, 4.0), (1.6, 0.05, 4.0), (1.8, 0.2, 4.0), (1.6, 0.1, 6.0), (1.8, 0.1, 6.0)):
    P = omc_amd.default_params(rho_scale=rs, slots=1024, relax=relax, rho_f_ratio=rf)
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    print("relax %.2f rho_f_ratio %.2f rho_scale %.1f: %.2fs, iterations total %d median %d, status %s" % (relax, rf, rs, el, it.sum(), np.median(it), st.tolist()), flush=True)
