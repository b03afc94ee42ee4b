"""Anderson acceleration on/off: iterations, statuses and batch time on a config-2 frontier."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P0 = omc_amd.default_params(rho_scale=4.0, accel=0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P0)
nodes = nodes[-256:]
res = {}
for name, kw in (("plain", dict(accel=0)), ("aa 10/5", dict(accel=1)), ("aa 10/5 nobump", dict(accel=1, bump_max=0)), ("aa 5/5", dict(accel=1, aa_mem=5)), ("aa 10/10", dict(accel=1, aa_every=10)), ("aa 10/5 start 25", dict(accel=1, aa_start=25))):
    P = omc_amd.default_params(rho_scale=4.0, slots=len(nodes), **kw)
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    obj = np.array([o["objective"] for o in out]); lb = np.array([o["dual_bound"] for o in out])
    res[name] = (obj, lb)
    ks = eng.kernel_stats()
    print("%-18s %.2fs status %s iters median %d mean %.0f max %d total %d | max rel gap %.1e | ms: %s" % (name, el, st, np.median(it), it.mean(), it.max(), it.sum(), ((obj - lb) / np.abs(obj)).max(),
          {k_: round(v["ms"]) for k_, v in ks.items()}), flush=True)
o0, l0 = res["plain"]; o1, l1 = res["aa 10/5"]
print("objective: max rel diff plain vs aa %.2e ; bounds consistent (lb_aa <= obj_plain + tol): %s ; (lb_plain <= obj_aa + tol): %s" % (
    np.max(np.abs(o0 - o1) / np.abs(o0)), bool((l1 <= o0 * (1 + 1e-9)).all()), bool((l0 <= o1 * (1 + 1e-9)).all())))
