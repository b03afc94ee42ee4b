"""A 100 x 100 instance on which B&B really branches (README type: A = randn, mask = Bernoulli(1/2)): frontier throughput."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
A, mask = omc_amd.pkg.data.readme_instance(100, 100, 0)
eng = omc_amd.Engine(A, mask, 80.0, 1)
rs, log = omc_amd.pkg.bnb.autotune_rho_scale(eng, "linear"); print("autotune", rs, log, flush=True)
P = omc_amd.default_params(rho_scale=rs)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 9
t0 = time.perf_counter(); nodes, levels = omc_amd.pkg.bnb.expand_frontier(eng, depth, "linear", params=P); print("frontier", len(nodes), "%.1fs" % (time.perf_counter() - t0), flush=True)
for acc, kw in ([(0, {}), (1, {}), (1, dict(aa_mem=5)), (1, dict(aa_mem=5, aa_every=10)), (1, dict(aa_mem=10, aa_every=10)), (1, dict(aa_mem=3))] if os.environ.get("ACCEL_TOO") else [(0, {})]):
  P = omc_amd.default_params(rho_scale=rs, slots=len(nodes), accel=acc, **kw)
  t0 = time.perf_counter()
  out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_Y=False, want_X=False)
  el = time.perf_counter() - t0
  print("accel", acc, kw, end=" ")
  it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
  lm = np.array([o["lambda_min"][0] for o in out])
  print("README-type 100x100: %d nodes in %.2fs = %.1f node-relaxations/s; status %s; iters median %d mean %.0f; lambda_min(UU'-Y) median %.3f; kernel ms %s" % (
      len(nodes), el, len(nodes) / el, st, np.median(it), it.mean(), np.median(lm), {k: round(v["ms"]) for k, v in eng.kernel_stats().items()}))
