"""k_colprox_pair diagnostics (OMC_SUB_DEBUG=4): inversions per column and the size of the first Halley step, config-2 depth-9 frontier warm-started."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OMC_SUB_DEBUG"] = "4"
import numpy as np, ctypes as C
import omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
depth = int(os.environ.get("DEPTH", "9"))
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
P = omc_amd.default_params(rho_scale=4.0, slots=1024)
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
lib = omc_amd.load()
for name, lf in (("warm", par), ("cold", None)):
    eng.stage(nodes, "linear", P, load_from=lf)
    o0 = np.zeros(32); lib.omc_debug_stamps(eng._h, o0.ctypes.data_as(C.c_void_p))
    eng.solve()
    o1 = np.zeros(32); lib.omc_debug_stamps(eng._h, o1.ctypes.data_as(C.c_void_p))
    h = o1 - o0
    tot = h[:8].sum()
    print(name, "columns", int(tot), "inversions per column", {i: round(float(v / tot), 4) for i, v in enumerate(h[:8]) if v}, "mean", round(float((h[:8] * np.arange(8)).sum() / tot), 3))
    print("   first step cp|d| ||z||/||y||, -log10 bins:", {i: round(float(v / tot), 4) for i, v in enumerate(h[8:24]) if v}, "guarded", int(h[24]), "first inversion failed", int(h[25]))
