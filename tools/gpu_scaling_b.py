"""Per-iteration time against the batch size on identical nodes (one depth-d node replicated B times): where does the engine stop being
latency-bound?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
nodes, _ = bnb.expand_frontier(eng, d, "linear", params=omc_amd.default_params(rho_scale=4.0, slots=256))
node = nodes[len(nodes) // 3]
for B in (1, 8, 32, 64, 128, 256, 512, 1024):
    P = omc_amd.default_params(rho_scale=4.0, slots=B, max_iters=300)
    eng.stage([node] * B, "linear", P); eng.solve()
    t0 = time.perf_counter(); eng.solve(); el = time.perf_counter() - t0
    out = eng.fetch(want_Y=False, want_X=False)
    its = max(o["iters"] for o in out)
    ks = eng.kernel_stats()
    print("B=%4d: %.0f us per iteration (%d iterations); per-launch us: %s" % (B, el / its * 1e6, its, {k: round(v["ms"] * 1e3 / max(1, v["launches"])) for k, v in ks.items() if v["launches"] > 0 and k in ("colprox", "cone_sub", "cone", "small", "global")}), flush=True)
