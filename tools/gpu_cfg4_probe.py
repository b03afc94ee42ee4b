"""BASELINE config 4 shape (500 x 500, rank 2, linear3, smallest_2_eigvec): how long does an ADMM iteration take now?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
A, mask, gamma, c = omc_amd.pkg.data.config_instance(cfg, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0, max_iters=iters, breakpoints=omc_amd.pkg.api.BREAKPOINTS[c["breakpoints"]])
t0 = time.perf_counter()
out = eng.matrix_completion_SDP_relaxation([[] for _ in range(B)], c["cut_type"], params=P, want_Y=False, want_X=False)
el = time.perf_counter() - t0
ks = eng.kernel_stats()
print("config %d: B=%d, %d iterations in %.2fs = %.1f ms / iteration; status %s obj %.6f lb %.6f" % (cfg, B, out[0]["iters"], el, el / max(1, out[0]["iters"]) * 1e3, out[0]["termination_status"], out[0]["objective"], out[0]["dual_bound"]))
print("kernel ms", {k: round(v["ms"], 1) for k, v in ks.items()}, "launches", {k: v["launches"] for k, v in ks.items()})
print("sub", eng.subspace_stats(), "info", eng.solver_info())
