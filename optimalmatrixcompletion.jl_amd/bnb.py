"""Host-side counterparts of the reference driver pieces that surround the hot path (OMC.jl = /root/reference/
src/OptimalMatrixCompletion.jl).  Only what is needed to feed and measure the HIP path lives here:

  child_directions / make_children   create_matrix_cut_child_nodes, bookkeeping part        OMC.jl:2479-2542
  expand_frontier                    breadth-first expansion, one GPU batch per level       (driver loop OMC.jl:700-1073)
  autotune_rho_scale                 picks the ADMM penalty once per instance at the root
  shard_nodes / allreduce_bounds     node-parallel sharding + the only exchange of B&B: min over ranks of
                                     {incumbent upper bound, smallest open lower bound}      
"""
from __future__ import annotations

import itertools

import numpy as np

DIRECTIONS_OF = {  # OMC.jl:2481-2491
    "linear": ["left", "right"],
    "linear2": ["left", "middle", "right"],
    "linear3": ["left", "inner_left", "inner_right", "right"],
}


def child_directions(cut_type, k):
    """Cartesian product of the direction labels with the FIRST column varying fastest (Iterators.product)."""
    labels = DIRECTIONS_OF[cut_type]
    return [list(c[::-1]) for c in itertools.product(*([labels] * k))]


def make_children(cuts, relax_result, cut_type, k):
    """Children of a node: parent's cut list + (breakpoint_vec, U_relax, directions)  (OMC.jl:2520-2542)."""
    x = np.array(relax_result["breakpoint_vec"]); U = np.array(relax_result["U"])
    return [list(cuts) + [(x, U, d)] for d in child_directions(cut_type, k)]


def autotune_rho_scale(engine, cut_type="linear", scales=(0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0), max_iters=600, return_result=False, **params):
    """Relax the root once per candidate penalty -- all candidates in ONE GPU batch (per-node penalties) -- and keep the
    scale that certifies the gap in the fewest iterations; the batch ends as soon as the first candidate is certified
    (`first_wins`).  Returns (scale, log) or, with return_result, (scale, log, root result of the winner)."""
    from .api import default_params
    p = default_params(rho_scale=1.0, max_iters=max_iters, first_wins=1, **params)
    outs = engine.matrix_completion_SDP_relaxation([[] for _ in scales], cut_type, params=p, rho_scales=list(scales))
    log = [(float(sc), o["status_code"], o["iters"]) for sc, o in zip(scales, outs)]
    ok = [(o["iters"], i) for i, o in enumerate(outs) if o["status_code"] == 0]
    if not ok:
        return (1.0, log, None) if return_result else (1.0, log)
    _, i = min(ok)
    return (float(scales[i]), log, outs[i]) if return_result else (float(scales[i]), log)


def expand_frontier(engine, depth, cut_type="linear", params=None, max_nodes=None):
    """Breadth-first expansion to `depth`: every level is one batched relaxation on the GPU.  Returns the list of
    node descriptors (cut lists) of the last level and the per-level results."""
    nodes = [[]]; levels = []
    for d in range(depth):
        out = engine.matrix_completion_SDP_relaxation(nodes, cut_type, params=params, want_Y=False, want_X=False)
        levels.append(out)
        new = []
        for cuts, o in zip(nodes, out):
            if not o["feasible"]:
                continue
            new.extend(make_children(cuts, o, cut_type, engine.k))
        nodes = new
        if max_nodes and len(nodes) >= max_nodes:
            nodes = nodes[:max_nodes]
            if d + 1 < depth:
                continue
    return nodes, levels


def shard_nodes(nodes, rank, world_size):
    """Round-robin in queue order so that every rank sees a similar mix of bounds."""
    return nodes[rank::world_size]


def allreduce_bounds(upper_bound, lower_bound, group=None):
    """min over ranks of {incumbent UB, smallest open LB}: 16 bytes, the only data-path exchange of node-parallel
    B&B.  Uses torch.distributed (backend nccl = RCCL on the GPU box, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(upper_bound), float(lower_bound)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(upper_bound), float(lower_bound)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t[0]), float(t[1])


class Comm:
    """The exchanges of node-parallel B&B, one process per GPU.  Per round: the per-node records every rank needs to
    grow the same tree (status, objective, bound, lambda_min, breakpoint vector, U: ~8 n (k + 1) bytes per node -- never X or Y),
    one MIN all-reduce of {incumbent UB, stop flag} and, only when the incumbent improved, a broadcast of X from its owner.
    The bound all-reduce and the X broadcast go through the engine's RCCL communicator (C ABI: omc_allreduce_bounds /
    omc_bcast_incumbent) when it has been initialised, otherwise through torch.distributed (gloo in the CPU tests)."""

    def __init__(self, rank=0, world_size=1, engine=None, group=None):
        self.rank, self.world, self.group = int(rank), int(world_size), group
        self.max_rows = 4096                 # rows per rank and round the record exchange is sized for (the driver pops at most `batch` nodes)
        self.engine = engine if (engine is not None and getattr(engine, "world_size", 1) == self.world and self.world > 1) else None
        if self.world > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group (rendezvous + record exchange)")
            self.dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def allgather_rows(self, rows, width):
        """rows: (cnt, width) float64 of this rank; returns the rows of all ranks in rank order."""
        rows = np.asarray(rows, dtype=np.float64).reshape(-1, width)
        if self.world == 1:
            return rows
        if self.engine is not None:          # the library's own RCCL communicator: no second transport beside it (omc_allgather_records)
            return self.engine.allgather_records(rows, width, self.max_rows * self.world)
        import torch
        import torch.distributed as dist
        cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=self.dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        cmax = int(max(int(c[0]) for c in cnts))
        pad = torch.zeros((max(cmax, 1), width), dtype=torch.float64, device=self.dev)
        if rows.shape[0]:
            pad[: rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows)).to(self.dev)
        parts = [torch.zeros_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        return np.concatenate([p[: int(c[0])].cpu().numpy() for p, c in zip(parts, cnts)], axis=0)

    def min_bounds(self, ub, lb):
        """MIN over ranks of (ub, lb) and the smallest rank that holds the minimal ub."""
        if self.world == 1:
            return float(ub), float(lb), 0
        if self.engine is not None:
            return self.engine.allreduce_bounds(ub, lb)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(ub), float(lb)], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        o = torch.tensor([float(self.rank) if float(ub) == float(t[0]) else float(self.world)], dtype=torch.float64, device=self.dev)
        dist.all_reduce(o, op=dist.ReduceOp.MIN, group=self.group)
        return float(t[0]), float(t[1]), int(o[0])

    def bcast_matrix(self, X, root):
        if self.world == 1:
            return X
        if self.engine is not None:
            return np.array(self.engine.bcast_incumbent(root, X))
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(self.dev)
        dist.broadcast(t, src=root, group=self.group)
        return t.cpu().numpy()

    def any_flag(self, flag):
        """True on every rank when it is true on one (loop termination must be a collective decision)."""
        if self.world == 1:
            return bool(flag)
        import torch
        import torch.distributed as dist
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t[0] > 0)


def left_singular(engine, X, k):
    """svd(X).U[:, 1:k] (OMC.jl:524, 564, 921) on the device (omc_left_singular_batch: Gram product on the matrix cores + the eigen-kernel);
    an engine without the entry point (a stub in the CPU tests) falls back to numpy."""
    if engine is not None and hasattr(engine, "left_singular"):
        return engine.left_singular([X])[0]
    return np.linalg.svd(X, full_matrices=False)[0][:, :k]


def rank_k_projection(X, k, engine=None):
    """X_k = U_k U_k' X with U_k = svd(X).U[:, 1:k] (OMC.jl:921): a matrix of rank <= k, so evaluate_objective(X_k) is a valid
    upper bound of the master problem whatever the relaxation's tolerances were."""
    Uf = left_singular(engine, X, k)
    return Uf @ (Uf.T @ X), Uf


# ----------------------------------------------------------------------------------------------------------
# Driver counterpart: matrix_completion_branchandbound (OMC.jl:140-1146) with the node evaluation on the GPU.
# ----------------------------------------------------------------------------------------------------------
def compute_gap(lower, upper):
    """OMC.jl:173-179."""
    return float("inf") if lower < 0 else upper / lower - 1.0


def branch_and_bound(engine, A, indices, *, node_selection="bestfirst", bestfirst_depthfirst_cutoff=10000, gap=1e-4,
                     disjunctive_cuts_type="linear", disjunctive_cuts_breakpoints="smallest_1_eigvec", root_only=False,
                     altmin_flag=True, max_altmin_probability=1.0, min_altmin_probability=0.005,
                     altmin_probability_decay_rate=1.1, use_max_steps=False, max_steps=1000000, time_limit=3600.0,
                     batch=64, rho_scale=None, params=None, seed=0, use_certified_bound=True, verbose=False,
                     rank=0, world_size=1, altmin_root_n_iters=1, accel=1,
                     add_Shor_valid_inequalities=False, Shor_valid_inequalities_noisy_rank1_num_entries_present=(1, 2, 3, 4),
                     add_Shor_valid_inequalities_fraction=1.0, add_Shor_valid_inequalities_iterative=False,
                     max_update_Shor_indices_probability=1.0, min_update_Shor_indices_probability=0.1,
                     update_Shor_indices_probability_decay_rate=1.1, update_Shor_indices_n_minors=100, shor_params=None,
                     warm_start=True, warm_pool_bytes=6 << 30):
    """Behavioural counterpart of the reference driver for use_disjunctive_cuts = true, no Shor, one altmin run at the
    root (altmin_root_n_iters = 1).  Differences, all deliberate: (1) up to `batch` nodes are popped per round in the
    reference's selection order and relaxed in ONE GPU batch (batch=1 reproduces the serial order); (2) the node bound
    is the certified dual bound (use_certified_bound) instead of the primal value of an eps-optimal point (quirk Q2);
    (3) the queue is a lazy-deletion heap instead of a rebuild per iteration (OMC.jl:1220-1244), same semantics:
    key = parent objective, ties by node id; (4) numpy's RNG replaces Julia's for the altmin coin flips (OMC.jl:867).
    With world_size > 1 every rank runs the same host logic on the same tree, relaxes its round-robin shard of the popped nodes
    and the ranks exchange (class Comm) the small per-node records, a MIN all-reduce of the incumbent and -- only when it improved --
    the incumbent X from its owner; leaving the loop is decided collectively.
    add_Shor_valid_inequalities (one rank; rank k > 1 through reference quirk Q5, see api.shor_rank_k_extension): every node carries node.Shor_info (OMC.jl:37-40).  Static mode (OMC.jl:646-669): the
    class lists of generate_rank1_matrix_completion_Shor_constraints_indexes (on the device), thinned by the fraction with numpy's RNG
    (randsubseq, OMC.jl:652-655); iterative mode (OMC.jl:670-674, 956-967, 2495-2518): the root starts without minors and a split node
    adds, with probability p(depth), the update_Shor_indices_n_minors most violated minors of its X (generate_violated_Shor_minors on
    the device) to the list its children inherit.  The SOC list is always "every coordinate outside the minors" (OMC.jl:656-665, 2508-2517).  (5) A master-feasible node updates the incumbent with
    evaluate_objective of the rank-k projection of its X, a certified value, instead of the relaxation objective (OMC.jl:818-828).
    Returns (solution, instance) dicts with the reference's key names where they apply (OMC.jl:604-621, 391-454)."""
    import heapq
    import time
    import math
    from .api import default_params, BREAKPOINTS
    if disjunctive_cuts_type not in DIRECTIONS_OF:
        raise ValueError("Invalid input for disjunctive cuts type.")                       # OMC.jl:217-224
    if disjunctive_cuts_breakpoints not in BREAKPOINTS:
        raise ValueError("Invalid input for disjunctive cuts breakpoints.")                # OMC.jl:225-231
    if node_selection not in ("breadthfirst", "bestfirst", "depthfirst", "bestfirst_depthfirst"):
        raise ValueError("Invalid input for node selection.")                              # OMC.jl:233-238
    n, m, k = engine.n, engine.m, engine.k
    if engine.k > 4 and altmin_flag:
        raise NotImplementedError("GPU altmin supports rank k <= 4; pass altmin_flag=False beyond that")
    rng = np.random.default_rng(seed)                                                      # OMC.jl:333 (Random.seed!(0))
    shor = bool(add_Shor_valid_inequalities)
    shor_classes = [int(c) for c in Shor_valid_inequalities_noisy_rank1_num_entries_present]
    if shor:
        if world_size != 1:
            raise NotImplementedError("Shor mode of the driver counterpart runs on one rank (the violated-minor update needs the node's X)")
        if not 0.0 <= add_Shor_valid_inequalities_fraction <= 1.0:
            raise ValueError("Argument `add_Shor_valid_inequalities_fraction` out of bounds [0.0, 1.0].")          # OMC.jl:256-263
        if add_Shor_valid_inequalities_iterative:
            if not 0.0 <= max_update_Shor_indices_probability <= 1.0:
                raise ValueError("Argument `max_update_Shor_indices_probability` out of bounds [0.0, 1.0].")       # OMC.jl:297-303
            if not 0.0 < min_update_Shor_indices_probability < 1.0:
                raise ValueError("Argument `min_update_Shor_indices_probability` out of bounds (0.0, 1.0).")       # OMC.jl:304-310
            if not 1.0 < update_Shor_indices_probability_decay_rate:
                raise ValueError("Argument `update_Shor_indices_probability_decay_rate` out of bounds (1.0, inf).")   # OMC.jl:311-317
            if not 1 <= update_Shor_indices_n_minors:
                raise ValueError("Argument `update_Shor_indices_n_minors` out of bounds [1.0, inf).")              # OMC.jl:318-324
    start = time.time()
    counters = dict(nodes_explored=0, nodes_total=1, nodes_dominated=0, nodes_relax_infeasible=0, nodes_relax_feasible=0,
                    nodes_relax_feasible_pruned=0, nodes_master_feasible=0, nodes_master_feasible_improvement=0,
                    nodes_relax_feasible_split=0, nodes_relax_feasible_split_altmin=0,
                    nodes_relax_feasible_split_altmin_improvement=0)
    run_log = []
    # ---- root altmin (OMC.jl:521-621) ------------------------------------------------------------------------
    A0 = np.where(indices, A, 0.0)
    U0 = left_singular(engine, A0, k)                                                      # OMC.jl:524
    solution = {}
    if altmin_flag:
        # OMC.jl:534-579: run 1 starts from the SVD of the zero-filled A, the others from it + sc * randn; all runs in ONE GPU
        # batch, the best objective wins
        sc = float(np.abs(U0).max())
        starts = [U0] + [U0 + sc * rng.standard_normal((n, k)) for _ in range(max(int(altmin_root_n_iters), 1) - 1)]
        ams = engine.alternating_minimization(starts, [[] for _ in starts], disjunctive_cuts_type)
        best = min(ams, key=lambda a: a["master_objective"])          # evaluate_objective(U V) from the factors, on the device
        X0 = best["U"] @ best["V"]
    else:
        X0 = U0 @ (U0.T @ A0)
    U_init = left_singular(engine, X0, k)                                                  # OMC.jl:564
    ub = float(engine.evaluate_objective(X0))
    from .data import compute_MSE
    mse0 = {kind: compute_MSE(X0, A, indices, kind) for kind in ("in", "out", "all")}                 # OMC.jl:570-572
    solution.update(objective_initial=ub, objective_initial_time_found=time.time() - start, X_initial=X0, U_initial=U_init,
                    Y_initial=U_init @ U_init.T, MSE_in_initial=mse0["in"], MSE_out_initial=mse0["out"], MSE_all_initial=mse0["all"],
                    objective=ub, X=X0, U=U_init, Y=U_init @ U_init.T, objective_time_found=time.time() - start)   # OMC.jl:604-621
    precomputed = {}
    if shor and rho_scale is None:
        rho_scale = 1.0                      # the Shor splitting has its own (scaled) penalties: no autotune
    if rho_scale is None:
        rho_scale, _, root_res = autotune_rho_scale(engine, disjunctive_cuts_type, return_result=True,
                                                    breakpoints=BREAKPOINTS[disjunctive_cuts_breakpoints])
        if root_res is not None and params is None:
            precomputed[1] = root_res                      # the winner of the autotune batch IS the root relaxation
    # Anderson acceleration is ON in the driver: on trees that really branch it halves the iterations per node (README instance: 263 -> 407
    # nodes/s, tools/gpu_bnb_cfg1.py); the library default stays off because the degenerate frontier of the bench instance rejects most points
    P = params or default_params(rho_scale=float(rho_scale), breakpoints=BREAKPOINTS[disjunctive_cuts_breakpoints], accel=int(accel))
    if shor:
        # 1e-5 is the tolerance SURVEY 8c states for the Shor configurations; the splitting needs a few thousand iterations per node
        PS = shor_params or default_params(rho_scale=1.0, breakpoints=BREAKPOINTS[disjunctive_cuts_breakpoints], eps_gap=1e-5, max_iters=8000)
        if add_Shor_valid_inequalities_iterative:
            root_minors = np.zeros((0, 4), np.int64)                                       # OMC.jl:670-674
        else:
            root_minors = np.asarray(engine.generate_rank1_matrix_completion_Shor_constraints_indexes(shor_classes), np.int64).reshape(-1, 4)
            if add_Shor_valid_inequalities_fraction < 1.0:                                 # randsubseq, OMC.jl:652-655
                root_minors = root_minors[rng.random(len(root_minors)) < add_Shor_valid_inequalities_fraction]
        shor_decay_depth = (math.log(max_update_Shor_indices_probability / min_update_Shor_indices_probability, update_Shor_indices_probability_decay_rate)
                            if add_Shor_valid_inequalities_iterative else 0.0)
    # ---- warm-start pool: ring of final states on the device, entry -> id of the node that owns it --------------------------------
    pool_cap = 0; pool_next = 0; pool_owner = {}
    if warm_start and not shor and hasattr(engine, "state_pool_create"):
        nnz = int(np.count_nonzero(indices)); np16 = (n + 15) // 16 * 16
        state_bytes = 8 * (3 * n * n + n * k + nnz + m + 16 * np16 + 20)
        pool_cap = int(max(0, min(1 << 17, warm_pool_bytes // state_bytes)))
        if pool_cap >= 2:
            engine.state_pool_create(pool_cap)
        else:
            pool_cap = 0
    # ---- tree ------------------------------------------------------------------------------------------------
    nodes = {1: dict(cuts=[], LB=-math.inf, depth=0, parent=0)}
    if shor:
        nodes[1]["shor"] = root_minors
    heap = [(math.inf, 1)]            # (key = parent objective, node id)   OMC.jl:697
    lbheap = [(-math.inf, 1)]         # (node bound, node id), lazy deletion: the global lower bound is its smallest live entry (OMC.jl:1207-1218)
    fifo = [1]
    ub_at_last_prune = math.inf
    lb = -math.inf
    now_gap = math.inf
    t_relax = t_altmin = 0.0
    altmin_decay_depth = math.log(max_altmin_probability / min_altmin_probability, altmin_probability_decay_rate) if altmin_flag else 0.0

    def pop_ids(cnt):
        out = []
        sel = node_selection
        if sel == "bestfirst_depthfirst":
            sel = "depthfirst" if len(nodes) > bestfirst_depthfirst_cutoff else "bestfirst"      # OMC.jl:709-717
        while len(out) < cnt and nodes:
            if sel == "bestfirst":
                while heap and heap[0][1] not in nodes:
                    heapq.heappop(heap)
                if not heap:
                    break
                _, nid = heapq.heappop(heap)
            elif sel == "breadthfirst":
                while fifo and fifo[0] not in nodes:
                    fifo.pop(0)
                if not fifo:
                    break
                nid = fifo.pop(0)
            else:
                while fifo and fifo[-1] not in nodes:
                    fifo.pop()
                if not fifo:
                    break
                nid = fifo.pop()
            if nid in nodes:
                out.append(nid)
        return out

    comm = Comm(rank, world_size, engine)
    REC = 6 + n + n * k              # record every rank needs per relaxed node: id, status, objective, bound, lambda_min[2], x, U

    def keep_going():
        stop = not (now_gap > gap and not (use_max_steps and counters["nodes_total"] >= max_steps) and time.time() - start <= time_limit and nodes)
        return not comm.any_flag(stop)           # the clocks of the ranks differ: leaving the loop is a collective decision

    while keep_going():
        ids = pop_ids(batch)
        if not ids:
            break
        popped = [(nid, nodes.pop(nid)) for nid in ids]
        counters["nodes_explored"] += len(popped)
        todo = []
        for nid, nd in popped:
            if nd["LB"] > ub:                                                               # OMC.jl:725-728
                counters["nodes_dominated"] += 1
            else:
                todo.append((nid, nd))
        if todo:
            t0 = time.time()
            mine = todo[rank::world_size] if world_size > 1 else todo                        # round-robin in queue order
            need = [(nid, nd) for nid, nd in mine if nid not in precomputed]
            if shor:
                fresh = engine.matrix_completion_SDP_relaxation([nd["cuts"] for _, nd in need], disjunctive_cuts_type, params=PS, want_X=True,
                                                                add_Shor_valid_inequalities=True,
                                                                shor_info=[(nd["shor"], None) for _, nd in need]) if need else []
            elif pool_cap and need:
                lf = []; sv = []
                for nid, nd in need:
                    ps = nd.get("pstate")          # (rank, pool entry, parent id): valid while the ring has not re-used the entry
                    lf.append(ps[1] if (ps is not None and ps[0] == rank and pool_owner.get(ps[1]) == ps[2]) else -1)
                busy = set(v for v in lf if v >= 0)          # entries this batch still reads: a slot may be set up after another one has been harvested
                for nid, nd in need:
                    while pool_next in busy and len(busy) < pool_cap:
                        pool_next = (pool_next + 1) % pool_cap
                    sv.append(pool_next); pool_owner[pool_next] = nid; pool_next = (pool_next + 1) % pool_cap
                counters["warm_started"] = counters.get("warm_started", 0) + sum(1 for v in lf if v >= 0)
                fresh = engine.matrix_completion_SDP_relaxation([nd["cuts"] for _, nd in need], disjunctive_cuts_type, params=P, want_X=True,
                                                                load_from=lf, save_to=sv)
                for (nid, nd), s_ in zip(need, sv):
                    nd["state"] = (rank, s_, nid)
            else:
                fresh = engine.matrix_completion_SDP_relaxation([nd["cuts"] for _, nd in need], disjunctive_cuts_type, params=P,
                                                                want_X=True) if need else []
            fresh = {nid: r for (nid, _), r in zip(need, fresh)}
            local = {nid: (precomputed.pop(nid) if nid in precomputed else fresh[nid]) for nid, _ in mine}
            # ---- exchange of the small per-node records (no X, no Y) ------------------------------------------------
            rows = np.zeros((len(mine), REC))
            for q, (nid, _) in enumerate(mine):
                r = local[nid]
                rows[q, :6] = (nid, r["status_code"], r["objective"], r["dual_bound"], r["lambda_min"][0], r["lambda_min"][1])
                rows[q, 6:6 + n] = r["breakpoint_vec"]; rows[q, 6 + n:] = np.asarray(r["U"]).ravel(order="F")
            allrows = comm.allgather_rows(rows, REC)
            rec = {int(row[0]): row for row in allrows}
            t_relax += time.time() - t0
            split = []; cand = []            # cand: (ub candidate, X) found by THIS rank in this round
            for nid, nd in todo:
                row = rec[nid]
                status, objective, bound_cert = int(row[1]), float(row[2]), float(row[3])
                if status == 3:                                                              # OMC.jl:777-779
                    counters["nodes_relax_infeasible"] += 1
                    continue
                counters["nodes_relax_feasible"] += 1                                       # OMC.jl:786
                bound = bound_cert if use_certified_bound else objective
                nd["LB"] = bound
                if nid == 1:
                    lb = bound                                                              # OMC.jl:793-795
                if bound > ub:                                                               # OMC.jl:797-800
                    counters["nodes_relax_feasible_pruned"] += 1
                    continue
                if status == 0 and row[4] >= -1e-6:                                          # OMC.jl:807-837
                    counters["nodes_master_feasible"] += 1
                    if nid in local:
                        # the reference takes the relaxation value as the incumbent (OMC.jl:818-828); a first-order solve certifies that
                        # value only to eps_gap, so the incumbent is the master objective of the rank-k projection of X instead
                        Xk, _ = rank_k_projection(local[nid]["X"], k, engine)
                        cand.append((float(engine.evaluate_objective(Xk)), Xk, "master"))
                    continue
                split.append((nid, nd, row))
            # altmin at split nodes w.p. p(depth)  (OMC.jl:856-949): every rank draws the same coins, runs its own share
            if altmin_flag and split:
                chosen = []
                for nid, nd, row in split:
                    p = min_altmin_probability if nd["depth"] > altmin_decay_depth else max_altmin_probability / (altmin_probability_decay_rate ** nd["depth"])
                    if rng.random() < p:
                        chosen.append((nid, nd))
                counters["nodes_relax_feasible_split_altmin"] += len(chosen)
                chosen = [(nid, nd) for nid, nd in chosen if nid in local]
                if chosen:
                    t0 = time.time()
                    Ur = engine.round_Y([local[nid]["Y"] for nid, _ in chosen])              # OMC.jl:873
                    ams = engine.alternating_minimization(Ur, [nd["cuts"] for _, nd in chosen], disjunctive_cuts_type)
                    # master objective of U V from the factors, on the device (OMC.jl:919-927): X is formed for the best one only
                    conv = [a for a in ams if a["converged"]]
                    if conv:
                        j = int(np.argmin([a["master_objective"] for a in conv]))
                        cand.append((conv[j]["master_objective"], conv[j]["U"] @ conv[j]["V"], "altmin"))
                    t_altmin += time.time() - t0
            # ---- incumbent: 16-byte MIN all-reduce; X travels only when the incumbent improved ---------------------------
            my_ub = min([c[0] for c in cand], default=math.inf)
            src = min(cand, key=lambda c: c[0])[2] if cand else "altmin"
            g_ub, _, owner = comm.min_bounds(my_ub, 0.0)
            # which kind of node found it: only ranks whose candidate IS the global minimum vote (ties: master); the others send 2
            _, g_src, _ = comm.min_bounds(my_ub, (0.0 if src == "master" else 1.0) if (cand and my_ub == g_ub) else 2.0)
            if g_ub < ub:
                Xl = min(cand, key=lambda c: c[0])[1] if (cand and comm.rank == owner) else np.zeros((n, m))
                Xl = comm.bcast_matrix(Xl, owner)
                ub = g_ub
                Ul = left_singular(engine, Xl, k)                                            # OMC.jl:921
                solution.update(objective=ub, X=Xl, U=Ul, Y=Ul @ Ul.T, objective_time_found=time.time() - start)
                counters["nodes_master_feasible_improvement" if g_src == 0.0 else "nodes_relax_feasible_split_altmin_improvement"] += 1
            for nid, nd, row in split:                                                       # OMC.jl:951-989
                counters["nodes_relax_feasible_split"] += 1
                r = dict(breakpoint_vec=row[6:6 + n].copy(), U=row[6 + n:].reshape((n, k), order="F").copy())
                kids = make_children(nd["cuts"], r, disjunctive_cuts_type, k)
                child_shor = nd.get("shor")
                if shor and add_Shor_valid_inequalities_iterative:                          # OMC.jl:956-967, 2495-2518
                    pu = (min_update_Shor_indices_probability if nd["depth"] > shor_decay_depth
                          else max_update_Shor_indices_probability / (update_Shor_indices_probability_decay_rate ** nd["depth"]))
                    if rng.random() < pu:
                        # OMC.jl:2497 passes reshape(X, (1, n, m)) whatever k is: the score is the minor of X itself (also what the k > 1 extension
                        # Xt_1 = X, Xt_t = 0 gives); the device routine takes (k, n, m)
                        X3 = np.zeros((k, n, m)); X3[0] = local[nid]["X"]
                        new = engine.generate_violated_Shor_minors(X3, shor_classes, [tuple(t) for t in child_shor.tolist()],
                                                                   int(update_Shor_indices_n_minors))
                        add = np.asarray([t for _, t in new], np.int64).reshape(-1, 4)
                        child_shor = np.concatenate([child_shor, add]) if len(add) else child_shor     # union (2504-2507): `new` excludes the existing ones
                        counters["shor_updates"] = counters.get("shor_updates", 0) + 1
                for cuts in kids:
                    counters["nodes_total"] += 1
                    cid = counters["nodes_total"]
                    nodes[cid] = dict(cuts=cuts, LB=nd["LB"], depth=nd["depth"] + 1, parent=nid)
                    if shor:
                        nodes[cid]["shor"] = child_shor
                    if nd.get("state") is not None:
                        nodes[cid]["pstate"] = nd["state"]
                    heapq.heappush(heap, (nd["LB"], cid)); heapq.heappush(lbheap, (nd["LB"], cid)); fifo.append(cid)
        # prune dominated nodes (OMC.jl:1220-1244): the open set can only gain dominated nodes when the incumbent improved, so the scan the
        # reference repeats every iteration runs only then; the global lower bound (OMC.jl:1207-1218) is the smallest live entry of a
        # lazy-deletion heap keyed by the node bounds (a child's bound is fixed when it is created)
        if ub < ub_at_last_prune:
            for nid in [i for i, nd in nodes.items() if nd["LB"] > ub]:
                del nodes[nid]
            ub_at_last_prune = ub
        while lbheap and (lbheap[0][1] not in nodes or nodes[lbheap[0][1]]["LB"] != lbheap[0][0]):
            heapq.heappop(lbheap)
        if nodes and lbheap:
            minval = lbheap[0][0]
            if minval > lb:
                lb = minval
        now_gap = compute_gap(lb, ub)
        run_log.append((counters["nodes_explored"], counters["nodes_total"], len(nodes), lb, ub, now_gap, time.time() - start))
        if verbose:
            print("| %10d | %10d | %10d | %10f | %10f | %10f | %10.3f  s  |" % run_log[-1], flush=True)
        if root_only:
            break
    elapsed = time.time() - start
    solution.update(lower_bound=lb, gap=now_gap, MSE_in=compute_MSE(solution["X"], A, indices, "in"),
                    MSE_out=compute_MSE(solution["X"], A, indices, "out"), MSE_all=compute_MSE(solution["X"], A, indices, "all"))   # OMC.jl:1079-1081
    # run_log columns as the reference's DataFrame (OMC.jl:457-465): explored, total, remaining, lower, upper, gap, runtime
    instance = dict(run_log=run_log, run_log_columns=("explored", "total", "remaining", "lower", "upper", "gap", "runtime"),
                    run_details=dict(counters, time_taken=elapsed, solve_time_relaxation=t_relax, solve_time_altmin=t_altmin,
                                                      rho_scale=float(rho_scale), batch=batch, n=n, m=m, k=k))
    return solution, instance
