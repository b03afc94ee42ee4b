import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, omc_amd
from test_golden import load, FILES
for f in FILES:
    z, nodes = load(f)
    eng = omc_amd.Engine(z["A"], z["mask"], float(z["gamma"]), int(z["k"]))
    out = eng.matrix_completion_SDP_relaxation(nodes, str(z["cut_type"]), params=omc_amd.default_params(rho_scale=float(z["rho_scale"])))
    print(os.path.basename(f))
    for b, o in enumerate(out):
        print("  node", b, "gpu obj %.6f lb %.6f st %d it %d | golden obj %.6f lb %.6f st %d it %d" % (o["objective"], o["dual_bound"], o["status_code"], o["iters"], z["objective"][b], z["dual_bound"][b], z["status"][b], z["iters"][b]))
    one = eng.matrix_completion_SDP_relaxation([nodes[0]], str(z["cut_type"]), params=omc_amd.default_params(rho_scale=float(z["rho_scale"])))[0]
    print("  node 0 alone: obj %.6f st %d it %d" % (one["objective"], one["status_code"], one["iters"]))
    eng.close()
