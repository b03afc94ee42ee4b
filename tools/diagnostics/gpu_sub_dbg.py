import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OMC_SUB_DEBUG"] = "1"; os.environ["OMC_STREAMS"] = "1"
import numpy as np, ctypes as C, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
P = omc_amd.default_params(rho_scale=4.0, slots=2048)
nodes, _ = bnb.expand_frontier(eng, 7, "linear", params=P)
for mi in (50, 100, 200, 400, 800):
    P2 = omc_amd.default_params(rho_scale=4.0, slots=2048, max_iters=mi)
    eng.stage(nodes, "linear", P2); eng.solve()
    out = np.zeros(32)
    omc_amd.load().omc_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p))
    print("max_iters", mi, eng.subspace_stats())
    print("  theta", np.array2string(np.sort(out[:16])[::-1], precision=3, max_line_width=250))
    print("  relres", np.array2string(out[16:][np.argsort(out[:16])[::-1]], precision=2, max_line_width=250))
eng.close()
