"""Batch-1 latency (BASELINE config 2 wording: single-node-at-a-time): microseconds per ADMM iteration, streams on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
P = omc_amd.default_params(rho_scale=4.0, slots=64)
nodes, _ = bnb.expand_frontier(eng, 5, "linear", params=P)
for B in (1, 8):
    P1 = omc_amd.default_params(rho_scale=4.0, slots=B)
    sample = [nodes[i:i + B] for i in range(0, 16, B)]
    eng.matrix_completion_SDP_relaxation(sample[0], "linear", params=P1, want_Y=False, want_X=False)
    t0 = time.perf_counter(); its = 0
    for c in sample:
        out = eng.matrix_completion_SDP_relaxation(c, "linear", params=P1, want_Y=False, want_X=False)
        its += max(o["iters"] for o in out)
    el = time.perf_counter() - t0
    print("OMC_STREAMS=%s B=%d: %.1f ms per call, %.0f us per iteration, kernel ms %s" % (os.environ.get("OMC_STREAMS", "-"), B, el / len(sample) * 1e3, el / its * 1e6,
          {k: round(v["ms"], 1) for k, v in eng.kernel_stats().items() if v["ms"] > 0}))
