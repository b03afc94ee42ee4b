"""Warm regime of the bench (every node started from its parent's state): which calls still reach the full eigen-kernel?  Histogram of the
number of positive eigenvalues of Y - D1 at those calls (OMC_SUB_DEBUG=3), plus the iteration counts of the nodes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OMC_SUB_DEBUG"] = "3"
import numpy as np, ctypes as C
import omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
depth = int(os.environ.get("DEPTH", "11"))
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
rs, _ = bnb.autotune_rho_scale(eng, "linear")
P = omc_amd.default_params(rho_scale=rs, slots=1024)
eng.state_pool_create(1 << (depth + 1))
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
eng.stage(nodes, "linear", P, load_from=par)
lib = omc_amd.load()
o0 = np.zeros(96); lib.omc_debug_stamps(eng._h, o0.ctypes.data_as(C.c_void_p))
eng.solve()
o1 = np.zeros(96); lib.omc_debug_stamps(eng._h, o1.ctypes.data_as(C.c_void_p))
h = o1 - o0
res = eng.fetch(want_Y=False, want_X=False)
it = np.array([x["iters"] for x in res]); st = np.bincount([x["status_code"] for x in res], minlength=4)
print("nodes", len(nodes), "status", st.tolist(), "iters median", int(np.median(it)), "mean", float(it.mean()), "sum", int(it.sum()), "full calls", int(h[:32].sum()), "sub", eng.subspace_stats())
print("iters histogram (bins of 100):", np.bincount(np.minimum(it // 100, 20)).tolist())
print("positive eigenvalues at full calls (31 = 31 or more):", {i: int(v) for i, v in enumerate(h[:32]) if v})
print("eigenvalues outside [0, 1]:", {i: int(v) for i, v in enumerate(h[32:64]) if v})
print("min of the two:", {i: int(v) for i, v in enumerate(h[64:96]) if v})
