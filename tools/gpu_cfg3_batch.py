"""BASELINE config 3 (200 x 200 rank 1, 20 % observed) as a batch: throughput of the L2-resident kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask, gamma, c = omc_amd.pkg.data.config_instance(3, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
t0 = time.perf_counter(); rs, log = omc_amd.pkg.bnb.autotune_rho_scale(eng, c["cut_type"]); print("autotune", rs, log, "%.1fs" % (time.perf_counter() - t0), flush=True)
P = omc_amd.default_params(rho_scale=rs)
t0 = time.perf_counter(); nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, int(sys.argv[1]) if len(sys.argv) > 1 else 7, c["cut_type"], params=P); print("frontier", len(nodes), "%.1fs" % (time.perf_counter() - t0), flush=True)
P = omc_amd.default_params(rho_scale=rs, slots=len(nodes))
t0 = time.perf_counter()
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
el = time.perf_counter() - t0
it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
print("config 3: %d nodes in %.2fs = %.1f node-relaxations/s; status %s iters median %d; kernel ms %s; info %s; sub %s" % (len(nodes), el, len(nodes) / el, st, np.median(it), {k: round(v["ms"]) for k, v in eng.kernel_stats().items()}, eng.solver_info(), eng.subspace_stats()))
