"""Round-based driver (bnb.branch_and_bound) vs the queue-driven one (bnb_stream.branch_and_bound_streaming) on the branching instance."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import omc_amd
bnb, bs, data = omc_amd.pkg.bnb, omc_amd.pkg.bnb_stream, omc_amd.pkg.data
TL = float(os.environ.get("TL", "30"))
OUT = open(os.path.join(ROOT, "gpurun_out", "bnb_stream.txt"), "w")
def say(*a):
    print(*a, flush=True); print(*a, file=OUT, flush=True)
A, mask = data.branching_instance(seed=0)
eng = omc_amd.Engine(A, mask, 80.0, 1)
for name in os.environ.get("RUNS", "round,stream").split(","):
    t0 = time.time()
    if name == "round":
        sol, inst = bnb.branch_and_bound(eng, A, mask, gap=1e-4, time_limit=TL, batch=256)
    else:
        sol, inst = bs.branch_and_bound_streaming(eng, A, mask, gap=1e-4, time_limit=TL, slots=int(os.environ.get("SLOTS", "1024")))
    rd = inst["run_details"]
    say(name, json.dumps(dict(seconds=round(time.time() - t0, 1), gap=sol["gap"], lower=sol["lower_bound"], upper=sol["objective"], explored=rd["nodes_explored"],
                              relaxed=rd["nodes_relax_feasible"], relax_s=round(rd["solve_time_relaxation"], 1), altmin_s=round(rd["solve_time_altmin"], 1),
                              nodes_per_s=round(rd["nodes_relax_feasible"] / max(rd["solve_time_relaxation"], 1e-9), 1), epochs=rd.get("epochs"), warm=rd.get("warm_started")), default=float))
