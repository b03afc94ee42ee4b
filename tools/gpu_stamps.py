import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omc_amd
from omc_amd_pkg import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib_stamps", "libomc_hip.so")
_lib._lib = None
A, mask, gamma, c = omc_amd.pkg.data.config_instance(2, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0, max_iters=400)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, int(sys.argv[1]) if len(sys.argv) > 1 else 4, c["cut_type"], params=P); nodes = nodes[:int(sys.argv[2]) if len(sys.argv) > 2 else 64]
out = eng.matrix_completion_SDP_relaxation(nodes, c["cut_type"], params=P, want_Y=False, want_X=False)
st = np.zeros(32); _lib.check(eng._lib.omc_debug_stamps(eng._h, _lib.ptr(st)))
its = out[0]["iters"]
names = {5: "cone: prologue", 6: "cone: zero fill", 0: "cone: gemm", 1: "cone: sweeps", 2: "cone: norms+Vrow", 3: "cone: select", 4: "cone: rebuild", 8: "glob: target", 9: "glob: LL scatter", 10: "glob: rows c", 11: "glob: nnqp", 12: "glob: U/V", 13: "glob: Y", 16: "small: T1", 17: "small: M3", 18: "small: eig", 19: "small: rebuild", 20: "small: E3", 21: "colprox(col 0): gather", 22: "colprox: L = B + cp s I", 23: "colprox: LDL", 24: "colprox: 2 solves", 25: "colprox: Taylor finish", 26: "colprox: loop tail", 27: "colprox: store"}
print("node0 iters", its, "s_memtime ticks = shader cycles (~2.4 GHz)")
for k_, nm in names.items(): print("%-20s %8.1f us per iteration" % (nm, st[k_] / 2400.0 / its))
