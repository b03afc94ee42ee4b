import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
DN = {0: "left", 1: "middle", 2: "right", 3: "inner_left", 4: "inner_right"}
for f in ("readme_20x24_k1_linear", "lowrank_24x28_k1_linear2", "lowrank_16x20_k2_linear3"):
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f + ".npz"), allow_pickle=False)
    cuts = [(z["cut_x"][l], z["cut_U"][l], [DN[int(c)] for c in z["cut_dir"][l]]) for l in range(len(z["cut_x"]))]
    nodes = [cuts[:int(L)] for L in z["node_L"]]
    eng = omc_amd.Engine(z["A"], z["mask"], 80.0, int(z["k"]))
    for acc in (0, 1):
        out = eng.matrix_completion_SDP_relaxation(nodes, str(z["cut_type"]), params=omc_amd.default_params(rho_scale=float(z["rho_scale"]), accel=acc), want_Y=False, want_X=False)
        print(f, "accel", acc, "iters", [o["iters"] for o in out], "status", [o["status_code"] for o in out], "golden iters", list(z["iters"]), "obj", ["%.7f" % o["objective"] for o in out])
    eng.close()
