"""Run the GPU-backed branch-and-bound on a synthetic instance and report nodes/s and wall-clock to gap."""
import sys, os, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omc_amd
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20); ap.add_argument("--m", type=int, default=25); ap.add_argument("--k", type=int, default=1)
ap.add_argument("--kind", default="readme"); ap.add_argument("--frac", type=float, default=0.3); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--gap", type=float, default=1e-4); ap.add_argument("--time-limit", type=float, default=300)
ap.add_argument("--cut-type", default="linear"); ap.add_argument("--verbose", action="store_true")
a = ap.parse_args()
data = omc_amd.pkg.data
A, mask = data.readme_instance(a.n, a.m, a.seed) if a.kind == "readme" else data.generate_matrix_completion_data(a.k, a.n, a.m, int(a.frac * a.n * a.m), a.seed)
eng = omc_amd.Engine(A, mask, 80.0, a.k)
sol, inst = omc_amd.pkg.bnb.branch_and_bound(eng, A, mask, gap=a.gap, time_limit=a.time_limit, batch=a.batch, disjunctive_cuts_type=a.cut_type, verbose=a.verbose)
d = inst["run_details"]
print(json.dumps(dict(n=a.n, m=a.m, k=a.k, kind=a.kind, batch=a.batch, upper=sol["objective"], lower=sol["lower_bound"], gap=sol["gap"],
                      time=d["time_taken"], relax_time=d["solve_time_relaxation"], altmin_time=d["solve_time_altmin"], nodes_relaxed=d["nodes_relax_feasible"],
                      nodes_per_s=d["nodes_relax_feasible"] / max(d["solve_time_relaxation"], 1e-9), counters={k_: v for k_, v in d.items() if k_.startswith("nodes_")}, rho_scale=d["rho_scale"])))
