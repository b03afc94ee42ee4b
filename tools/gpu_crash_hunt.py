"""Hunt for the sporadic host crash: the round-based driver on a tiny instance (constant re-capture of the iteration graph in the draining tails,
OMC_GRAPH_TAILS=1), engines created and closed in a loop, native frames on a fatal signal (OMC_SEGV_TRACE=1)."""
import os, sys, time, faulthandler
faulthandler.enable()
os.environ.setdefault("OMC_SEGV_TRACE", "1"); os.environ.setdefault("OMC_GRAPH_TAILS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd, omc_oracle as orc
bnb = omc_amd.pkg.bnb
N = int(os.environ.get("REPS", "30")); TL = float(os.environ.get("TL", "4"))
A, mask = orc.make_instance(14, 18, 1, seed=5, kind="lowrank", n_indices=int(0.35 * 14 * 18), noise=0.15)
for i in range(N):
    eng = omc_amd.Engine(A, mask, 80.0, 1)
    s, inst = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=TL, batch=64, rho_scale=8.0)
    print(i, inst["run_details"]["nodes_explored"], flush=True)
    eng.close()
print("clean", flush=True)
