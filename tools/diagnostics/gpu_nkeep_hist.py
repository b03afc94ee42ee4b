"""Histogram of the number of positive eigenvalues of the cone input Y - D1 at the calls that reach the full eigendecomposition
(k_cone_ws) -- the calls the tracked 16-vector block cannot take over while more than 12 eigenvalues are positive.  OMC_SUB_DEBUG=3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["OMC_SUB_DEBUG"] = "3"
import numpy as np, ctypes as C, omc_amd
cfgi, depth = int(sys.argv[1]), int(sys.argv[2])
A, mask, gamma, c = omc_amd.pkg.data.config_instance(cfgi, seed=0)
eng = omc_amd.Engine(A, mask, gamma, c["k"])
P = omc_amd.default_params(rho_scale=4.0)
nodes, _ = omc_amd.pkg.bnb.expand_frontier(eng, depth, c["cut_type"], params=P)
z = np.zeros(32); 
P = omc_amd.default_params(rho_scale=4.0, slots=len(nodes))
eng.stage(nodes, c["cut_type"], P)
out0 = np.zeros(96); omc_amd.load().omc_debug_stamps(eng._h, out0.ctypes.data_as(C.c_void_p))
eng.solve()
out = np.zeros(96); omc_amd.load().omc_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p))
h = out - out0
print("config", cfgi, "nodes", len(nodes), "full calls", int(h[:32].sum()), "sub", eng.subspace_stats())
print("positive eigenvalues (31 = 31 or more):", {i: int(v) for i, v in enumerate(h[:32]) if v})
print("eigenvalues outside [0, 1] (the defect side):", {i: int(v) for i, v in enumerate(h[32:64]) if v})
print("min of the two (what a two-sided block would have to hold):", {i: int(v) for i, v in enumerate(h[64:96]) if v})
