"""Per-slot cost spread (diagnostic build, lib_stamps): how far is each launch's duration (max over workgroups) from the
mean cost per node?  One batch of <= slots nodes so that slot == node."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omc_amd
from omc_amd_pkg import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "lib_stamps", "libomc_hip.so")
_lib._lib = None
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
nodes = nodes[-256:]
P = omc_amd.default_params(rho_scale=4.0, slots=len(nodes))
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
S = len(nodes)
d = np.zeros(8 * S); _lib.check(eng._lib.omc_debug_diag(eng._h, _lib.ptr(d))); d = d.reshape(8, S)
its = np.array([o["iters"] for o in out], dtype=float)
st = np.array([o["status_code"] for o in out])
ks = eng.kernel_stats()
print("nodes", S, "status counts", np.bincount(st, minlength=4), "iters median/mean/max", np.median(its), its.mean(), its.max())
print("kernel stats", ks)
m = c["m"] if "m" in c else A.shape[1]
us = lambda cyc: cyc / 2400.0
cp = us(d[0]) / its / m          # per column per iteration (sum over waves / columns)
nf = d[1] / its / m
cone = us(d[2]) / np.maximum(d[3], 1)
glob = us(d[4]) / its
small = us(d[5]) / its
for nm, v in (("colprox us/column-wave", cp), ("factorizations/column", nf), ("cone us/call", cone), ("global us/call", glob), ("small us/call", small)):
    print("%-26s mean %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f | OPT mean %8.2f  SLOW mean %8.2f" % (
        nm, v.mean(), np.median(v), np.percentile(v, 90), v.max(), v[st == 0].mean() if (st == 0).any() else np.nan, v[st == 1].mean() if (st == 1).any() else np.nan))
print("Taylor finish audit: max rel err of alpha over all nodes %.3e (median of per-node max %.3e); max |phi(s)|/s %.3e" % (d[6].max(), np.median(d[6]), d[7].max()))
