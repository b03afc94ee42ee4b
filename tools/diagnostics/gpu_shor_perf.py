"""Throughput of the Shor-minor kernels (rows a10 / a11) against the HBM roofline.
Algorithmic bytes: 32 B per emitted tuple (a10); 16 B per candidate key written + 16 B per key and radix pass read (a11)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omc_amd
out = []
for cfg, classes, do_idx in ((3, [4], True), (3, [4, 3], True), (5, [4], False)):
    A, mask, gamma, c = omc_amd.pkg.data.config_instance(cfg, seed=0)
    eng = omc_amd.Engine(A, mask, gamma, c["k"])
    t0 = time.perf_counter(); cnt = eng.shor_count(classes); t_count = time.perf_counter() - t0
    rec = {"config": cfg, "n": c["n"], "m": c["m"], "k": c["k"], "classes": classes, "counts": [int(v) for v in cnt], "count_wall_s": t_count}
    if do_idx:
        eng.generate_rank1_matrix_completion_Shor_constraints_indexes(classes)            # warm-up (allocations)
        t0 = time.perf_counter(); T = eng.generate_rank1_matrix_completion_Shor_constraints_indexes(classes); wall = time.perf_counter() - t0
        st = eng.shor_last_stats()
        rec["indexes"] = {"tuples": len(T), "device_ms": st["ms"], "wall_s_incl_d2h": wall, "GBps_write": 32.0 * len(T) / (st["ms"] * 1e-3) / 1e9}
    rng = np.random.default_rng(0)
    X3 = np.stack([A + 0.1 * rng.standard_normal(A.shape) for _ in range(c["k"])])
    eng.generate_violated_Shor_minors(X3, classes, [], 100)
    t0 = time.perf_counter(); top = eng.generate_violated_Shor_minors(X3, classes, [], 100); wall = time.perf_counter() - t0
    st = eng.shor_last_stats()
    rec["violated"] = {"candidates": st["candidates"], "device_ms": st["ms"], "wall_s": wall, "Mcand_per_s": st["candidates"] / (st["ms"] * 1e-3) / 1e6,
                       "GBps_key_write_only": 16.0 * st["candidates"] / (st["ms"] * 1e-3) / 1e9, "top_score": top[0][0] if top else None}
    print(json.dumps(rec), flush=True)
    eng.close()
