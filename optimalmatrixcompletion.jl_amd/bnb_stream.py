"""Queue-driven counterpart of the reference's branch-and-bound loop (OMC.jl:700-1073) on ONE running solve.

`bnb.branch_and_bound` works in rounds: pop a batch, relax it to its last node, branch.  Here the engine's solve stays open
(omc_relax_hold), the host takes the results of the nodes that have finished (omc_relax_fetch_done), prunes / updates the incumbent /
creates the children exactly as the reference's loop does for one node (OMC.jl:765-1031), and pushes the best open nodes into the
running solve (omc_relax_append) so that the slots never drain while the queue holds work.  Same selection rule (best-first on the
parent's bound), same altmin coin (OMC.jl:856-870), same certified bounds as `bnb.branch_and_bound`; what differs is only WHEN a node
is relaxed relative to its cousins (several hundred nodes are in flight, as with batch > 1 there).  Rank one process, disjunctive cuts
only (the Shor lists and the multi-rank exchange stay with the round-based driver).  A second handle of the same instance serves
altmin, rounding and the objective scans while the first one is busy with the solve.
"""
from __future__ import annotations

import heapq
import math
import time

import numpy as np

from .bnb import make_children, autotune_rho_scale, compute_gap, left_singular


def _incumbent_from_U(engine2, A, indices, U, gamma):
    """Best X = U V for a fixed column space (one V-step of OMC.jl:1979-2279, closed form per column):
    v_j = argmin 1/2 sum_{i in Omega_j} (A_ij - U_i v)^2 + 1/(2 gamma) ||U v||^2."""
    n, m = A.shape
    k = U.shape[1]
    G0 = (U.T @ U) / gamma
    V = np.zeros((k, m))
    for j in range(m):
        o = indices[:, j]
        Uo = U[o]
        V[:, j] = np.linalg.solve(Uo.T @ Uo + G0 + 1e-14 * np.eye(k), Uo.T @ A[o, j])
    X = U @ V
    return float(engine2.evaluate_objective(X)), X


def branch_and_bound_streaming(engine, A, indices, *, gap=1e-4, time_limit=3600.0, disjunctive_cuts_type="linear",
                               disjunctive_cuts_breakpoints="smallest_1_eigvec", rho_scale=None, altmin_flag=True,
                               max_altmin_probability=1.0, min_altmin_probability=0.005, altmin_probability_decay_rate=1.1,
                               slots=1024, in_flight_target=None, capacity=1 << 15, depth_reserve=8, seed=0, accel=1,
                               warm_pool_bytes=6 << 30, verbose=False):
    from .api import default_params, BREAKPOINTS, Engine
    from .data import compute_MSE
    n, m, k = engine.n, engine.m, engine.k
    A = np.asarray(A, float); indices = np.asarray(indices, bool)
    rng = np.random.default_rng(seed)
    start = time.time()
    engine2 = Engine(A, indices, engine.gamma, k, device=getattr(engine, "device", 0))
    engine.tuning_set("OMC_NO_GRAPH", "1")      # two handles work side by side: no stream capture on the solving one while the other issues copies and launches
    counters = dict(nodes_explored=0, nodes_total=1, nodes_dominated=0, nodes_relax_infeasible=0, nodes_relax_feasible=0,
                    nodes_relax_feasible_pruned=0, nodes_master_feasible=0, nodes_master_feasible_improvement=0,
                    nodes_relax_feasible_split=0, nodes_relax_feasible_split_altmin=0, nodes_relax_feasible_split_altmin_improvement=0,
                    warm_started=0, epochs=0)
    # ---- root altmin (OMC.jl:521-621) on the second handle ----------------------------------------------------------------
    A0 = np.where(indices, A, 0.0)
    U0 = left_singular(engine2, A0, k)
    solution = {}
    if altmin_flag:
        am = engine2.alternating_minimization([U0], [[]], disjunctive_cuts_type)[0]
        X0 = am["U"] @ am["V"]
    else:
        X0 = U0 @ (U0.T @ A0)
    ub = float(engine2.evaluate_objective(X0))
    solution.update(objective_initial=ub, X_initial=X0, objective=ub, X=X0, U=left_singular(engine2, X0, k))
    # ---- penalty scale and root relaxation (the winner of the autotune batch IS the root relaxation) --------------------------
    if rho_scale is None:
        rho_scale, _, root = autotune_rho_scale(engine, disjunctive_cuts_type, return_result=True, breakpoints=BREAKPOINTS[disjunctive_cuts_breakpoints])
    else:
        root = None
    P = default_params(rho_scale=float(rho_scale), breakpoints=BREAKPOINTS[disjunctive_cuts_breakpoints], accel=int(accel), slots=int(slots))
    if root is None:
        root = engine.matrix_completion_SDP_relaxation([[]], disjunctive_cuts_type, params=P, want_X=False)[0]
    # ---- warm-start pool: a ring much longer than the in-flight window, entry -> node that owns it -----------------------------
    nnz = int(np.count_nonzero(indices)); np16 = (n + 15) // 16 * 16
    state_bytes = 8 * (3 * n * n + n * k + nnz + m + 16 * np16 + 20)
    target = int(in_flight_target or 2 * slots)
    pool_cap = int(max(0, min(1 << 16, warm_pool_bytes // state_bytes)))
    if pool_cap >= 8 * target:
        engine.state_pool_create(pool_cap)
    else:
        pool_cap = 0
    pool_next = 0; pool_owner = {}
    altmin_decay_depth = math.log(max_altmin_probability / min_altmin_probability, altmin_probability_decay_rate) if altmin_flag else 0.0

    heap = []          # (LB, id): open nodes, best-first on the parent's certified bound (OMC.jl:1164-1182)
    nodes = {}         # id -> dict(cuts, LB, depth, pstate)
    lb = -math.inf
    t_altmin = 0.0
    pending_altmin = []          # (Y, cuts) of split nodes that won the altmin coin: relaxed in batches on the second handle

    def consider(cand_val, cand_X, kind):
        nonlocal ub
        if cand_val < ub:
            ub = cand_val
            solution.update(objective=ub, X=cand_X, U=left_singular(engine2, cand_X, k), objective_time_found=time.time() - start)
            counters["nodes_master_feasible_improvement" if kind == "master" else "nodes_relax_feasible_split_altmin_improvement"] += 1
            for nid in [i for i, nd in nodes.items() if nd["LB"] > ub]:       # prune dominated nodes (OMC.jl:1220-1244)
                del nodes[nid]

    def process(nid, nd, o):
        """One node's share of OMC.jl:765-1031."""
        nonlocal lb
        counters["nodes_explored"] += 1
        if o["status_code"] == 3:
            counters["nodes_relax_infeasible"] += 1
            return
        counters["nodes_relax_feasible"] += 1
        bound = o["dual_bound"]
        if nid == 1:
            lb = bound
        if bound > ub:
            counters["nodes_relax_feasible_pruned"] += 1
            return
        if o["status_code"] == 0 and o["lambda_min"][0] >= -1e-6:
            counters["nodes_master_feasible"] += 1
            val, Xk = _incumbent_from_U(engine2, A, indices, o["U"], engine.gamma)
            consider(val, Xk, "master")
            return
        counters["nodes_relax_feasible_split"] += 1
        if altmin_flag and "Y" in o:
            p = min_altmin_probability if nd["depth"] > altmin_decay_depth else max_altmin_probability / (altmin_probability_decay_rate ** nd["depth"])
            if rng.random() < p:
                counters["nodes_relax_feasible_split_altmin"] += 1
                pending_altmin.append((o["Y"], nd["cuts"]))
        for cuts in make_children(nd["cuts"], o, disjunctive_cuts_type, k):
            counters["nodes_total"] += 1
            cid = counters["nodes_total"]
            nodes[cid] = dict(cuts=cuts, LB=bound, depth=nd["depth"] + 1, pstate=nd.get("state"))
            heapq.heappush(heap, (bound, cid))

    def run_altmin(force=False):
        nonlocal t_altmin
        if not pending_altmin or (len(pending_altmin) < 8 and not force):
            return
        t0 = time.time()
        batch = pending_altmin[:64]; del pending_altmin[:64]
        Ur = engine2.round_Y([y for y, _ in batch])                                           # OMC.jl:873
        ams = [a for a in engine2.alternating_minimization(Ur, [c for _, c in batch], disjunctive_cuts_type) if a["converged"]]
        if ams:
            best = min(ams, key=lambda a: a["master_objective"])
            consider(best["master_objective"], best["U"] @ best["V"], "altmin")
        t_altmin += time.time() - t0

    def pop_best(limit, depth_cap):
        out = []
        while heap and len(out) < limit:
            _, nid = heap[0]
            if nid not in nodes:
                heapq.heappop(heap); continue
            if nodes[nid]["LB"] > ub:
                heapq.heappop(heap); del nodes[nid]; counters["nodes_dominated"] += 1; continue
            if nodes[nid]["depth"] > depth_cap:
                break
            heapq.heappop(heap)
            out.append((nid, nodes.pop(nid)))
        return out

    def warm_indices(batch):
        nonlocal pool_next
        if not pool_cap:
            return None, None
        lf = []; sv = []
        for nid, nd in batch:
            ps = nd.get("pstate")
            ok = ps is not None and pool_owner.get(ps[0]) == ps[1]
            lf.append(ps[0] if ok else -1)
            counters["warm_started"] += 1 if ok else 0
        for nid, nd in batch:
            sv.append(pool_next); pool_owner[pool_next] = nid; nd["state"] = (pool_next, nid); pool_next = (pool_next + 1) % pool_cap
        return lf, sv

    # root: already relaxed; its state is not in the pool (children of the root start cold)
    process(1, dict(cuts=[], LB=-math.inf, depth=0), dict(root, Y=root.get("Y")) if root.get("Y") is not None else root)
    run_altmin(force=True)
    now_gap = compute_gap(lb, ub)
    run_log = [(counters["nodes_explored"], counters["nodes_total"], len(nodes), lb, ub, now_gap, time.time() - start)]
    relax_seconds = 0.0

    def open_lb(in_flight):
        while heap and heap[0][1] not in nodes:
            heapq.heappop(heap)
        vals = [heap[0][0]] if heap else []
        vals += [nd["LB"] for _, nd in in_flight.values()]
        return min(vals) if vals else None

    while now_gap > gap and time.time() - start <= time_limit and nodes:
        # ---- one epoch = one running solve; a new one is staged when the tree outgrows the reserved cut depth or the node capacity ----------
        counters["epochs"] += 1
        best = [nid for _, nid in heapq.nsmallest(slots, heap) if nid in nodes]          # the nodes this epoch starts with decide the cut depth it reserves
        depth_cap = (max(nodes[nid]["depth"] for nid in best) if best else 0) + depth_reserve
        first = pop_best(slots, depth_cap)
        if not first:
            break
        lf, sv = warm_indices(first)
        P.time_limit = max(1.0, time_limit - (time.time() - start))
        engine.reserve(capacity, depth_cap)
        engine.stage([nd["cuts"] for _, nd in first], disjunctive_cuts_type, P, load_from=lf, save_to=sv)
        engine.hold(True)
        t_epoch = time.time()
        engine.submit()
        in_flight = {i: first[i] for i in range(len(first))}
        sent = len(first); stop_push = False
        while True:
            got = engine.fetch_done(max_nodes=1024, want_Y=altmin_flag)
            for o in got:
                nid, nd = in_flight.pop(o["node"])
                process(nid, nd, o)
            if got:
                run_altmin()
                v = open_lb(in_flight)
                if v is not None and v > lb:
                    lb = v
                now_gap = compute_gap(lb, ub)
                run_log.append((counters["nodes_explored"], counters["nodes_total"], len(nodes) + len(in_flight), lb, ub, now_gap, time.time() - start))
                if verbose:
                    print("| %10d | %10d | %10d | %10f | %10f | %10f | %10.3f  s  |" % run_log[-1], flush=True)
            done = now_gap <= gap or time.time() - start > time_limit or (not nodes and not in_flight)
            if done:
                break
            if not stop_push and len(in_flight) < target:
                room = min(target - len(in_flight), capacity - (sent - len(first)))
                push = pop_best(room, depth_cap) if room > 0 else []
                if push:
                    lf, sv = warm_indices(push)
                    try:
                        engine.append([nd["cuts"] for _, nd in push], disjunctive_cuts_type, load_from=lf, save_to=sv)
                    except Exception:                              # the solve ended on its time limit between our checks: back into the queue
                        for nid, nd in push:
                            nodes[nid] = nd; heapq.heappush(heap, (nd["LB"], nid))
                        stop_push = True
                        push = []
                    for q, item in enumerate(push):
                        in_flight[sent + q] = item
                    sent += len(push)
                elif room <= 0 or (heap and nodes):
                    stop_push = True            # capacity used up, or the best open node lies below the reserved depth: drain and stage a new solve
            if stop_push and not in_flight:
                break
            if not got:
                time.sleep(0.0005)
        engine.hold(False)
        engine.wait()
        for o in engine.fetch_done(max_nodes=1 << 15, want_Y=False):      # nodes that finished after the loop decided to stop: their bounds still count
            if o["node"] in in_flight:
                nid, nd = in_flight.pop(o["node"])
                process(nid, nd, o)
        for nid, nd in in_flight.values():              # (none unless the time limit struck) -- back into the queue
            nodes[nid] = nd; heapq.heappush(heap, (nd["LB"], nid))
        relax_seconds += time.time() - t_epoch
        run_altmin(force=True)
        v = open_lb({})
        if v is not None and v > lb:
            lb = v
        now_gap = compute_gap(lb, ub)
    elapsed = time.time() - start
    solution.update(lower_bound=lb, gap=now_gap, MSE_in=compute_MSE(solution["X"], A, indices, "in"),
                    MSE_out=compute_MSE(solution["X"], A, indices, "out"), MSE_all=compute_MSE(solution["X"], A, indices, "all"))
    engine2.close()
    engine.tuning_set("OMC_NO_GRAPH", None)
    instance = dict(run_log=run_log, run_log_columns=("explored", "total", "remaining", "lower", "upper", "gap", "runtime"),
                    run_details=dict(counters, time_taken=elapsed, solve_time_relaxation=relax_seconds, solve_time_altmin=t_altmin,
                                     rho_scale=float(rho_scale), slots=slots, in_flight_target=target, n=n, m=m, k=k))
    return solution, instance
