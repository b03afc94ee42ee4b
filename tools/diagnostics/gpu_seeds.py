"""Robustness: default solver on frontiers of other config-2 seeds and of config 1 (README type)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
for cfg, seed, depth in ((2, 1, 9), (2, 2, 9), (2, 3, 9), (1, 0, 8)):
    A, mask, gamma, c = omc_amd.pkg.data.config_instance(cfg, seed=seed)
    eng = omc_amd.Engine(A, mask, gamma, c["k"])
    rs, _ = omc_amd.pkg.bnb.autotune_rho_scale(eng, c["cut_type"])
    P = omc_amd.default_params(rho_scale=rs)
    nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
    P = omc_amd.default_params(rho_scale=rs, slots=len(nodes))
    t0 = time.perf_counter()
    out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
    el = time.perf_counter() - t0
    it = np.array([o["iters"] for o in out]); st = np.bincount([o["status_code"] for o in out], minlength=4)
    obj = np.array([o["objective"] for o in out]); lb = np.array([o["dual_bound"] for o in out])
    ok = np.isfinite(obj[st[None, :].argmax() >= 0]).all()
    feas = np.array([o["feasible"] for o in out])
    print("config %d seed %d: rho_scale %.2f, %d nodes in %.2fs (%.0f/s) status %s iters median %d; finite %s; lb<=obj violations %d; worst gap %.1e" % (
        cfg, seed, rs, len(nodes), el, len(nodes) / el, st, np.median(it), bool(np.isfinite(obj[feas]).all() and np.isfinite(lb[feas]).all()),
        int((lb[feas] > obj[feas] * (1 + 1e-6) + 1e-9).sum()), ((obj[feas] - lb[feas]) / np.abs(obj[feas])).max()), flush=True)
    eng.close()
