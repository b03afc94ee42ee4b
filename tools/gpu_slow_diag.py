"""Which nodes of the config-2 frontier come back SLOW_PROGRESS, and what do they look like (development diagnostic)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd, omc_oracle as orc
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
depth = int(os.environ.get("DEPTH", "8"))
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
rs, _ = bnb.autotune_rho_scale(eng, "linear")
P = omc_amd.default_params(rho_scale=rs, slots=1024)
nodes, _ = bnb.expand_frontier(eng, depth, "linear", params=P)
out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_X=False)
rp, rd = np.zeros(len(nodes)), np.zeros(len(nodes))
import ctypes
eng._lib.omc_debug_residuals(eng._h, rp.ctypes.data_as(ctypes.c_void_p), rd.ctypes.data_as(ctypes.c_void_p))
rows = []
for i, (cuts, o) in enumerate(zip(nodes, out)):
    ev = np.linalg.eigvalsh(o["Y"])
    widths = []; slack = []
    for (x, Uh, dirs) in cuts:
        vhat = float(Uh[:, 0] @ x); v = float(o["U"][:, 0] @ x)
        lo, hi, sl, ic = orc.cut_piece("linear", dirs[0], vhat)
        widths.append(hi - lo); slack.append(min(v - lo, hi - v))
        # aggregated row slack: g(v) - x'Yx
        slack.append(sl * v + ic - float(x @ o["Y"] @ x))
    rows.append(dict(i=i, st=o["status_code"], it=o["iters"], gap=(o["objective"] - o["dual_bound"]) / abs(o["objective"]), rp=rp[i], rd=rd[i], npos=int((ev > 1e-6).sum()), ev_top=ev[-3:].round(4).tolist(),
                     lam_min=float(o["lambda_min"][0]), min_width=min(widths), min_slack=min(slack), n_tight=int(sum(1 for s_ in slack if abs(s_) < 1e-6))))
slow = [r for r in rows if r["st"] == 1]; ok = [r for r in rows if r["st"] == 0]
print("nodes", len(rows), "slow", len(slow))
def summ(rs_, name):
    if not rs_: return
    print(name, "iters med", np.median([r["it"] for r in rs_]), "npos med", np.median([r["npos"] for r in rs_]), "min_width med", np.median([r["min_width"] for r in rs_]), "n_tight med", np.median([r["n_tight"] for r in rs_]),
          "lam_min med", np.median([r["lam_min"] for r in rs_]), "gap med", np.median([r["gap"] for r in rs_]), "rp/rd med", np.median([r["rp"] / max(r["rd"], 1e-300) for r in rs_]))
summ(ok, "OPTIMAL"); summ(slow, "SLOW")
for r in slow[:12]: print(json.dumps(r))
