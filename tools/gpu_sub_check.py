"""Subspace-tracking cone kernel (k_cone_sub) against the full eigendecomposition path on config 2: same iteration counts / objectives,
kernel times, and how often the tracked subspace had to fall back."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
A, mask, gamma, cfg = data.config_instance(cfg_id, seed=0)
res = {}
for mode in ("full", "sub"):
    if mode == "full": os.environ["OMC_NO_SUBSPACE"] = "1"
    else: os.environ.pop("OMC_NO_SUBSPACE", None)
    eng = omc_amd.Engine(A, mask, gamma, cfg["k"])
    P = omc_amd.default_params(rho_scale=4.0, slots=2048)
    if mode == "full":
        nodes, _ = bnb.expand_frontier(eng, depth, cfg["cut_type"], params=P)
    eng.stage(nodes, cfg["cut_type"], P)
    eng.solve()
    t0 = time.perf_counter(); eng.solve(); el = time.perf_counter() - t0
    out = eng.fetch(want_Y=False, want_X=False)
    ks = eng.kernel_stats()
    res[mode] = dict(seconds=el, iters=[o["iters"] for o in out], obj=[o["objective"] for o in out], lb=[o["dual_bound"] for o in out], st=[o["status_code"] for o in out],
                     kernel_ms={k: round(v["ms"], 1) for k, v in ks.items()}, launches={k: v["launches"] for k, v in ks.items()}, sub=eng.subspace_stats())
    print(mode, "nodes", len(nodes), "seconds %.3f" % el, "nodes/s %.1f" % (len(nodes) / el), "kernel_ms", res[mode]["kernel_ms"], "sub", res[mode]["sub"], flush=True)
    eng.close()
a, b = res["full"], res["sub"]
it_a, it_b = np.array(a["iters"]), np.array(b["iters"])
print("iters equal on %d of %d nodes; max |diff| %d; status equal %d" % ((it_a == it_b).sum(), len(it_a), np.abs(it_a - it_b).max(), (np.array(a["st"]) == np.array(b["st"])).sum()))
rel = np.abs(np.array(a["obj"]) - np.array(b["obj"])) / np.maximum(1, np.abs(np.array(a["obj"])))
print("max rel objective diff %.2e ; max rel bound diff %.2e" % (rel.max(), (np.abs(np.array(a["lb"]) - np.array(b["lb"])) / np.maximum(1, np.abs(np.array(a["lb"])))).max()))
