"""Penalty scan of the Shor splitting at 100 x 100 (config-2 instance, all class-4 minors): gap after a fixed number of iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
data = omc_amd.pkg.data
n = int(os.environ.get("N", "100")); IT = int(os.environ.get("ITERS", "3000"))
import importlib
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import omc_oracle as orc
A, mask = orc.make_instance(n, n, 1, n_indices=int(0.2 * n * n), seed=0, noise=0.01)
eng = omc_amd.Engine(A, mask, 80.0, 1)
minors = eng.generate_rank1_matrix_completion_Shor_constraints_indexes([4])
print("minors", len(minors), flush=True)
scales = [0.25, 1.0, 4.0, 16.0]
for (r4, r5) in eval(os.environ.get("GRID", "[(20, 2), (5, 2), (80, 2), (20, 0.5), (20, 8), (300, 2)]")):
    p = omc_amd.default_params(max_iters=IT, eps_gap=1e-5, slots=len(scales))
    import ctypes
    eng._lib.omc_set_node_rho_scales(eng._h, len(scales), np.array(scales).ctypes.data_as(ctypes.c_void_p))
    t0 = time.time()
    eng.stage_shor([[]] * len(scales), [(minors, None)] * len(scales), "linear", p, penalties=(0.05, r4, r5))
    eng.solve(); out = eng.fetch(want_Y=False, want_X=False)
    print((r4, r5), " ".join(f"rho={0.05*s_:g}:{o['iters']}{'*' if o['status_code']==0 else ''}/{(o['objective']-o['dual_bound'])/abs(o['objective']):.1e}" for s_, o in zip(scales, out)), f"{time.time()-t0:.0f}s", flush=True)
