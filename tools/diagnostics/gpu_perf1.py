"""Config-2 style performance probe: expand a tree breadth-first on the GPU, time the widest level."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import omc_amd
import omc_oracle as orc

n = m = int(os.environ.get("N", 100)); k = 1; gamma = 80.0
depth = int(os.environ.get("DEPTH", 6))
A, mask = orc.make_instance(n, m, k, seed=0, kind="lowrank", n_indices=int(0.2 * n * m))
eng = omc_amd.Engine(A, mask, gamma, k)
# rho autotune at the root
best = None
for sc in [0.25, 0.5, 1, 2, 4, 8]:
    t = time.time(); out = eng.matrix_completion_SDP_relaxation([[]], "linear", params=omc_amd.default_params(rho_scale=sc, max_iters=3000), want_X=False)
    o = out[0]; print("root scale", sc, o["termination_status"], o["iters"], "obj %.8f lb %.8f" % (o["objective"], o["dual_bound"]), "%.2fs" % (time.time() - t), flush=True)
    if o["status_code"] == 0 and (best is None or o["iters"] < best[1]): best = (sc, o["iters"])
sc = best[0]; print("chosen rho_scale", sc)
P = omc_amd.default_params(rho_scale=sc, max_iters=3000)
nodes = [[]]
for d in range(depth + 1):
    t = time.time(); eng.stage(nodes, "linear", P); ts = time.time() - t
    t = time.time(); eng.solve(); tsol = time.time() - t
    t = time.time(); out = eng.fetch(want_Y=False, want_X=False); tf = time.time() - t
    its = np.array([o["iters"] for o in out]); st = np.array([o["status_code"] for o in out])
    objs = np.array([o["objective"] for o in out])
    print(f"depth {d}: B={len(nodes)} stage {ts:.3f}s solve {tsol:.3f}s fetch {tf:.3f}s  nodes/s {len(nodes)/tsol:.1f}  iters min/med/max {its.min()}/{int(np.median(its))}/{its.max()}  status {np.bincount(st, minlength=4)}  obj {objs.min():.5f}..{objs.max():.5f}", flush=True)
    ks = eng.kernel_stats(); info = eng.solver_info()
    print("    ", {c: (v["launches"], round(v["ms"], 1)) for c, v in ks.items()}, "sweeps/eig %.2f" % (info["jacobi_sweeps"] / max(1, ks["cone"]["units"] + ks["check"]["units"])), info)
    if d == depth: break
    new = []
    for c, o in zip(nodes, out):
        if o["status_code"] == 3: continue
        for dr in ["left", "right"]:
            new.append(c + [(o["breakpoint_vec"], o["U"], [dr])])
    nodes = new
eng.close()
