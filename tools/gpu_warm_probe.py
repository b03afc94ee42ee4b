"""Warm start from the parent's state vs cold start: iterations, full eigendecompositions and wall time on a config-2 frontier."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
depth = int(os.environ.get("DEPTH", "8"))
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
rs, _ = bnb.autotune_rho_scale(eng, "linear")
P = omc_amd.default_params(rho_scale=rs, slots=1024)
eng.state_pool_create(1 << (depth + 1))
# breadth-first expansion keeping every node's state: node i at level d has pool entry base_d + i
nodes = [[]]; sid = [0]; nxt = 1
out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_Y=False, want_X=False, save_to=sid)
for d in range(depth):
    kids = []; par = []
    for cuts, o, s_ in zip(nodes, out, sid):
        if not o["feasible"]:
            continue
        for c in bnb.make_children(cuts, o, "linear", 1):
            kids.append(c); par.append(s_)
    nodes = kids
    if d == depth - 1:
        break
    sid = list(range(nxt, nxt + len(nodes))); nxt += len(nodes)
    out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_Y=False, want_X=False, load_from=par, save_to=sid)
res = {}
for name, lf in (("cold", None), ("warm", par), ("cold2", None), ("warm2", par)):
    eng.stage(nodes, "linear", P, load_from=lf)
    t0 = time.time(); eng.solve(); t = time.time() - t0
    o = eng.fetch(want_Y=False, want_X=False)
    it = np.array([x["iters"] for x in o]); st = np.bincount([x["status_code"] for x in o], minlength=4)
    res[name] = dict(nodes=len(nodes), seconds=round(t, 3), nodes_per_s=round(len(nodes) / t, 1), certified_per_s=round((st[0] + st[3]) / t, 1), iters_median=int(np.median(it)), iters_mean=float(it.mean()),
                     status=st.tolist(), sub=eng.subspace_stats(), obj=[x["objective"] for x in o])
d1 = np.array(res["cold"]["obj"]); d2 = np.array(res["warm"]["obj"])
ok = [(a, b) for a, b, x, y in zip(d1, d2, res["cold"]["status"], res["warm"]["status"])]
print("max rel objective difference (all nodes):", float(np.max(np.abs(d1 - d2) / np.abs(d1))))
for k_, v in res.items():
    v.pop("obj"); print(k_, json.dumps(v))
