"""Wall time of a streamed frontier (1024 nodes through 512 slots, config 2) as a function of the check / refill period."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P0 = omc_amd.default_params(rho_scale=4.0, slots=1024)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 10, c["cut_type"], params=P0)
for ce in (25, 50, 25, 50, 40):
    P = omc_amd.default_params(rho_scale=4.0, slots=512, check_every=ce)
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    print("check_every %d: %.2fs = %.0f nodes/s, iterations total %d median %d, status %s" % (ce, el, len(nodes) / el, it.sum(), np.median(it), st.tolist()), flush=True)
