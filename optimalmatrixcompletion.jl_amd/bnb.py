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


def autotune_rho_scale(engine, cut_type="linear", scales=(0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0), max_iters=3000, **params):
    """Solve the root once per candidate and keep the penalty scale that certifies the gap in the fewest iterations."""
    from .api import default_params
    best = None; log = []
    for sc in scales:
        p = default_params(rho_scale=float(sc), max_iters=max_iters, **params)
        out = engine.matrix_completion_SDP_relaxation([[]], cut_type, params=p, want_Y=False, want_X=False)[0]
        log.append((float(sc), out["status_code"], out["iters"]))
        if out["status_code"] == 0 and (best is None or out["iters"] < best[1]):
            best = (float(sc), out["iters"])
    return (best[0] if best else 1.0), log


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
