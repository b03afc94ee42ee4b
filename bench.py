"""bench.py -- node-relaxations/sec of the HIP hot path on BASELINE config 2 (100x100, k=1, gamma=80, 20% observed,
linear cuts, smallest_1_eigvec).

A "step" = one pass of the hot path over one batch: B independent B&B nodes (a breadth-first frontier of the
config-2 tree, staged in HBM before the timed region) are relaxed to a certified gap by omc_relax_solve.
value = B * steps * n_gpus / max-over-ranks wall time.  Multi-GPU: one process per GPU, every rank relaxes its own
batch of the same size (weak scaling; independent nodes, no data-path collective) and the ranks exchange
min{incumbent UB, open LB} with an RCCL all-reduce after every step -- the one real exchange of node-parallel B&B.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_PEAK_TFLOPS = 78.6   # AMD MI355X datasheet (vector = matrix fp64); the local guide lists no fp64 peak


def f_proj(N):
    """algorithmic flops of one spectral projection of an order-N symmetric matrix."""
    return 13.0 / 3.0 * N ** 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--depth", type=int, default=int(os.environ.get("OMC_BENCH_DEPTH", 11)), help="frontier depth: 2^depth nodes per GPU per step")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("OMC_BENCH_SLOTS", 2048)), help="nodes relaxed concurrently per GPU (continuous batching)")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--accel", type=int, default=int(os.environ.get("OMC_BENCH_ACCEL", 0)), help="1: Anderson acceleration of the ADMM map (library default 0)")
    ap.add_argument("--cpu-nodes", type=int, default=2, help="nodes relaxed by the CPU oracle for cpu_baseline (rank 0, N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import omc_amd
    bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data

    A, mask, gamma, cfg = data.config_instance(args.config, seed=0)
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    eng = omc_amd.Engine(A, mask, gamma, k, device=local)
    rho_scale, tune_log = bnb.autotune_rho_scale(eng, cfg["cut_type"])
    P = omc_amd.default_params(rho_scale=rho_scale, slots=args.slots, accel=args.accel)
    # every rank builds the same frontier (deterministic) and keeps a shard-sized batch: rank r takes a rotated copy so
    # that ranks do not all hold the identical node order
    nodes, _ = bnb.expand_frontier(eng, args.depth, cfg["cut_type"], params=P)
    B = len(nodes)
    nodes = nodes[rank % B:] + nodes[:rank % B]
    eng.stage(nodes, cfg["cut_type"], P)           # node descriptors resident in HBM before the timed region

    def step():
        eng.solve()
        out = eng.fetch(want_Y=False, want_X=False)
        ub = min(o["objective"] for o in out); lb = min(o["dual_bound"] for o in out)
        return bnb.allreduce_bounds(ub, lb), out

    for _ in range(args.warmup):
        step()
    kstats = {}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, out = step()
        for c, v in eng.kernel_stats().items():
            a = kstats.setdefault(c, dict(launches=0, ms=0.0, units=0))
            a["launches"] += v["launches"]; a["ms"] += v["ms"]; a["units"] += v["units"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])
    info = eng.solver_info()
    value = B * args.steps * world / el
    iters = np.array([o["iters"] for o in out]); status = np.bincount([o["status_code"] for o in out], minlength=4)

    # roofline of the dominant kernel (k_cone: batched order-n spectral projection, fp64 compute bound):
    # algorithmic flops per launch = F_proj(n) x matrices in the launch; duration from HIP events on the solver's stream
    cone = kstats["cone"]
    mats_per_launch = cone["units"] / max(1, cone["launches"])
    avg_ms = cone["ms"] / max(1, cone["launches"])
    achieved = f_proj(n) * mats_per_launch / (avg_ms * 1e-3) / 1e12
    roofline = dict(bound="mfma", kernel="k_cone", achieved=achieved, peak=F64_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / F64_PEAK_TFLOPS,
                    traffic=None, avg_launch_ms=avg_ms, matrices_per_launch=mats_per_launch, order=n,
                    concurrency="k_cone_ws shares the CUs with k_colprox and k_small of the same iteration (three HIP streams), so its launch "
                                "duration includes that sharing; OMC_STREAMS=1 runs the kernels back to back (DESIGN.md 5.1 has both)",
                    kernel_ms={c: round(v["ms"], 2) for c, v in kstats.items()})

    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_nodes > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import omc_oracle as orc
        inst = orc.Instance(A, mask, gamma, k)
        sample = nodes[: args.cpu_nodes]
        t1 = time.perf_counter()
        for c in sample:
            orc.sdp_relaxation(inst, c, cfg["cut_type"], params=orc.RelaxParams(rho_scale=rho_scale), want_certificate=False)
        tc = time.perf_counter() - t1
        cpu_baseline = dict(value=len(sample) / tc, unit="node-relaxations/s", cores=1, kind="port",
                            sample=f"first {len(sample)} nodes of the same depth-{args.depth} frontier, numpy/LAPACK oracle, 1 thread-equivalent; "
                                   "the reference itself (Julia+Mosek) cannot run here")
    time_to_gap = None
    if rank == 0 and world == 1:
        # second half of the metric: wall-clock of the whole B&B (root altmin, penalty autotune, tree) to gap <= 1e-4
        tt = []
        for sd in (0, 1, 2):
            A2, mask2, g2, c2 = data.config_instance(args.config, seed=sd)
            e2 = omc_amd.Engine(A2, mask2, g2, c2["k"], device=local)
            t1 = time.perf_counter()
            sol, inst2 = bnb.branch_and_bound(e2, A2, mask2, gap=1e-4, time_limit=120.0, batch=128, disjunctive_cuts_type=c2["cut_type"])
            tt.append(dict(seed=sd, seconds=time.perf_counter() - t1, gap=sol["gap"], nodes_relaxed=inst2["run_details"]["nodes_relax_feasible"]))
            e2.close()
        time_to_gap = dict(target_gap=1e-4, runs=tt, median_seconds=float(np.median([t_["seconds"] for t_ in tt])))
    if rank == 0:
        print(json.dumps({
            "metric": "B&B node-relaxations/sec, 100x100 k=1", "value": value, "unit": "node-relaxations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {n}x{m} rank-{k}, gamma=80, 20% observed, {cfg['cut_type']} cuts, smallest_1_eigvec; "
                                   f"{B} depth-{args.depth} frontier nodes per GPU per step through {min(args.slots, B)} slots (continuous batching)", "nodes_per_gpu": B, "slots": min(args.slots, B), "rho_scale": rho_scale,
                       "eps_gap": 1e-6, "iters_median": int(np.median(iters)), "iters_max": int(iters.max()),
                       "status_counts": {"optimal": int(status[0]), "slow_progress": int(status[1]), "time_limit": int(status[2]), "infeasible": int(status[3])},
                       "jacobi_sweeps_last_step": info["jacobi_sweeps"], "instance_sha256": data.instance_sha256(A, mask)[:16]},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "time_to_gap": time_to_gap,
        }))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
