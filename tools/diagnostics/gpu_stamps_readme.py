import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
from omc_amd_pkg import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "lib_stamps", "libomc_hip.so"); _lib._lib = None
A, mask = omc_amd.pkg.data.readme_instance(100, 100, 0)
print("column 0 holds", int(mask[:, 0].sum()), "observed entries; max", int(mask.sum(0).max()))
eng = omc_amd.Engine(A, mask, 80.0, 1)
P = omc_amd.default_params(rho_scale=32.0, max_iters=300)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, 4, "linear", params=P)
out = eng.matrix_completion_SDP_relaxation(nodes, "linear", params=P, want_Y=False, want_X=False)
st = np.zeros(32); _lib.check(eng._lib.omc_debug_stamps(eng._h, _lib.ptr(st)))
its = out[0]["iters"]
names = {21: "colprox(col 0): gather", 22: "colprox: L = B + cp s I", 23: "colprox: LDL", 24: "colprox: 2 solves", 25: "colprox: Taylor finish", 26: "colprox: loop tail", 27: "colprox: store"}
print("node0 iters", its)
for k_, nm in names.items(): print("%-26s %8.1f us per iteration" % (nm, st[k_] / 2400.0 / its))
S = len(nodes); d = np.zeros(8 * S); _lib.check(eng._lib.omc_debug_diag(eng._h, _lib.ptr(d))); d = d.reshape(8, S)
print("factorizations per column-call (node 0): %.2f" % (d[1][0] / its / 100))
