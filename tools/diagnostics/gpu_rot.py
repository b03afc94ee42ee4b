"""Diagnostic build: rotations per sweep index and sweeps-per-call histogram of k_cone_ws for the node in slot 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, omc_amd
from omc_amd_pkg import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "lib_stamps", "libomc_hip.so"); _lib._lib = None
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
for first in (0, 5, 17, 40):
    sel = nodes[first:first + 16]           # slots >= 8 so that the diag rows 6, 7 of slot 0..7 exist
    out = eng.matrix_completion_SDP_relaxation(sel, c["cut_type"], params=omc_amd.default_params(rho_scale=4.0, slots=len(sel)), want_Y=False, want_X=False)
    S = len(sel); d = np.zeros(8 * S); _lib.check(eng._lib.omc_debug_diag(eng._h, _lib.ptr(d))); d = d.reshape(8, S)
    it = out[0]["iters"]
    print("node", first, "iters", it, "status", out[0]["status_code"], "| rotations per call by sweep index:", np.round(d[6][:8] / it, 1), "| calls ending after s sweeps:", d[7][:8].astype(int))
