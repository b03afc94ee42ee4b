"""BASELINE config 3 in Shor mode (200 x 200 rank 1, static class-4 list from the device enumeration): time per ADMM iteration and per
kernel class for a small batch, with the tracked subspace of the big cone on / off (OMC_SHOR_NO_SUBSPACE=1)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd
import omc_oracle as orc
B = int(os.environ.get("B", "8")); IT = int(os.environ.get("ITERS", "300")); frac = float(os.environ.get("FRAC", "1.0"))
os.environ["OMC_GRAPH_MAX"] = "0"          # eager launches: per-kernel HIP events
A, mask = orc.make_instance(200, 200, 1, n_indices=8000, seed=0, noise=0.01)
eng = omc_amd.Engine(A, mask, 80.0, 1)
t0 = time.time(); minors = eng.generate_rank1_matrix_completion_Shor_constraints_indexes([4]); t_enum = time.time() - t0
if frac < 1.0:
    minors = minors[np.random.default_rng(0).random(len(minors)) < frac]
p = omc_amd.default_params(max_iters=IT, slots=B, eps_gap=1e-5)
t0 = time.time(); eng.stage_shor([[]] * B, [(minors, None)] * B, "linear", p); t_stage = time.time() - t0
t0 = time.time(); eng.solve(); t_solve = time.time() - t0
out = eng.fetch(want_Y=False, want_X=False)
ks = {k_: (round(v["ms"], 1), v["launches"]) for k_, v in eng.kernel_stats().items() if v["launches"]}
print(json.dumps(dict(minors=int(len(minors)), B=B, iters=int(out[0]["iters"]), enum_s=round(t_enum, 3), stage_s=round(t_stage, 3), solve_s=round(t_solve, 3),
                      ms_per_iteration=round(t_solve / max(1, out[0]["iters"]) * 1e3, 3), node_iterations_per_s=round(B * out[0]["iters"] / t_solve, 1),
                      objective=out[0]["objective"], dual_bound=out[0]["dual_bound"], status=out[0]["status_code"], kernels_ms_launches=ks,
                      no_subspace=bool(os.environ.get("OMC_SHOR_NO_SUBSPACE")), big_sub=eng.shor_subspace_stats(), sub=eng.subspace_stats())))
