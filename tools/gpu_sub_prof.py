import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OMC_SUB_DEBUG"] = "2"; os.environ["OMC_STREAMS"] = "1"
import numpy as np, ctypes as C, omc_amd
bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
A, mask, gamma, cfg = data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, 1)
nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = omc_amd.default_params(rho_scale=4.0, slots=2048)
nodes, _ = bnb.expand_frontier(eng, int(np.log2(nslots)), "linear", params=P)
eng.stage(nodes, "linear", P); eng.solve()
out = np.zeros(32)
omc_amd.load().omc_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p))
names = ["load", "shift", "mul_MX", "gram", "chol+inv", "apply", "RR mul+gram", "eig16", "rotate", "resid", "store X", "rebuild W1"]
calls = max(out[12], 1)
print("slot 0: calls %d, steps/call %.2f, RR/call %.2f" % (out[12], out[13] / calls, out[14] / calls))
tot = out[:12].sum()
for nm, v in zip(names, out[:12]):
    print("  %-12s %9.0f cycles/call  %5.1f %%" % (nm, v / calls, 100 * v / tot))
print("  total %.0f cycles/call = %.1f us at 100 MHz ticks?  (s_memtime unit per guide: shader cycle)" % (tot / calls, tot / calls / 100.0))
print(eng.subspace_stats(), eng.kernel_stats()["cone_sub"])
