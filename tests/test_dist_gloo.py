"""world_size-2 gloo test of the only multi-GPU exchange of node-parallel B&B: sharding of independent nodes and the
min all-reduce of {incumbent upper bound, smallest open lower bound}.  Runs on CPU."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import omc_amd
    bnb = omc_amd.pkg.bnb
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nodes = list(range(11))
    mine = bnb.shard_nodes(nodes, rank, world)
    # pretend each node produced (upper, lower) = (100 - id, id): the global result must be the min over ALL nodes
    ub = min(100.0 - i for i in mine); lb = min(float(i) for i in mine)
    gub, glb = bnb.allreduce_bounds(ub, lb)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, gub, glb))


def test_sharding_and_bound_allreduce_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    res.sort()
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    for _, _, gub, glb in res:
        assert gub == 90.0 and glb == 0.0


# ---- the driver counterpart itself on two ranks -------------------------------------------------------------------------
class OracleEngine:
    """Test double with the Engine interface that bnb.branch_and_bound uses, backed by the CPU oracle (the HIP path needs a GPU;
    this test is about the exchange logic of the ranks, not about the solver)."""

    def __init__(self, A, mask, gamma, k):
        import omc_oracle as orc
        self.orc = orc; self.inst = orc.Instance(A, mask, gamma, k)
        self.n, self.m = A.shape; self.k = k; self.A, self.mask, self.gamma = A, mask, gamma
        self.calls = 0

    def matrix_completion_SDP_relaxation(self, nodes, cut_type="linear", params=None, want_X=True, **kw):
        orc = self.orc; out = []
        for cuts in nodes:
            self.calls += 1
            r = orc.sdp_relaxation(self.inst, cuts, cut_type, params=orc.RelaxParams(rho_scale=params.rho_scale, max_iters=600), want_certificate=False)
            x, ev = orc.breakpoint_vector(r["Y"], r["U"])
            ev = list(ev) + [0.0]
            out.append(dict(objective=r["objective"], dual_bound=r["dual_bound"], status_code=r["termination_status"], feasible=r["feasible"],
                            iters=r["iters"], U=r["U"], Y=r["Y"], X=r["X"], lambda_min=[ev[0], ev[1]], breakpoint_vec=x))
        return out

    def round_Y(self, Ys):
        return [self.orc.svd_rounding(Y, self.k) for Y in Ys]

    def alternating_minimization(self, U0s, nodes=None, cut_type="linear", **kw):
        out = []
        for u0, cuts in zip(U0s, nodes):
            r = self.orc.alternating_minimization(self.inst, u0, cuts, cut_type)
            r["master_objective"] = self.orc.evaluate_objective(r["U"] @ r["V"], self.A, self.mask, self.gamma)
            out.append(r)
        return out

    def evaluate_objective(self, X):
        import numpy as np
        X = np.asarray(X)
        if X.ndim == 2:
            return self.orc.evaluate_objective(X, self.A, self.mask, self.gamma)
        return np.array([self.orc.evaluate_objective(x, self.A, self.mask, self.gamma) for x in X])


def _bnb_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    import numpy as np
    import torch.distributed as dist
    import omc_amd, omc_oracle as orc
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    A, mask = orc.make_instance(10, 12, 1, seed=11, kind="readme")
    eng = OracleEngine(A, mask, 80.0, 1)
    sol, inst = omc_amd.pkg.bnb.branch_and_bound(eng, A, mask, gap=1e-3, time_limit=600.0, batch=4, rho_scale=16.0, use_max_steps=True, max_steps=24,
                                                 rank=rank, world_size=world, accel=0)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    q.put((rank, world, sol["lower_bound"], sol["objective"], inst["run_details"]["nodes_explored"], inst["run_details"]["nodes_relax_feasible"], eng.calls,
           float(np.abs(sol["X"]).sum()), [tuple(r[:6]) for r in inst["run_log"]]))


def test_branch_and_bound_two_ranks_matches_single_rank():
    """bnb.branch_and_bound on two gloo ranks (nodes sharded round-robin, records all-gathered, incumbent all-reduced + X broadcast from
    its owner, collective loop exit) must grow the same tree and end with the same [LB, UB] as the single-rank run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=_bnb_worker, args=(0, 1, port, q))] + [ctx.Process(target=_bnb_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=900) for _ in ps]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    single = [r for r in res if r[1] == 1][0]; two = sorted(r for r in res if r[1] == 2)
    for r in two:
        assert r[2] == single[2] and r[3] == single[3]                 # identical doubles: the same relaxations, the same order of decisions
        assert r[4] == single[4] and r[5] == single[5] and r[7] == single[7]
        assert [t[:6] for t in r[8]] == [t[:6] for t in single[8]]     # explored / total / remaining / lower / upper / gap per round
    assert two[0][6] + two[1][6] == single[6] and min(two[0][6], two[1][6]) > 0      # the relaxations were shared, none was repeated


# ---- Comm through an engine-side communicator (the C-ABI path: omc_allreduce_bounds / omc_bcast_incumbent / omc_allgather_records) ------
class StubCommEngine:
    """Stands in for the library's RCCL communicator with the same three calls and the same semantics (owner = smallest rank that holds the
    minimal upper bound; broadcast from any root; rows of every rank in rank order), carried by gloo so that it runs without a GPU."""

    def __init__(self, rank, world):
        self.rank, self.world_size = rank, world
        self.calls = dict(allreduce=0, bcast=0, allgather=0)

    def allreduce_bounds(self, ub, lb):
        import torch, torch.distributed as dist
        self.calls["allreduce"] += 1
        t = torch.tensor([float(ub), float(lb)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        o = torch.tensor([float(self.rank) if float(ub) == float(t[0]) else float(self.world_size)], dtype=torch.float64)
        dist.all_reduce(o, op=dist.ReduceOp.MIN)
        return float(t[0]), float(t[1]), int(o[0])

    def bcast_incumbent(self, root, X):
        import numpy as np, torch, torch.distributed as dist
        self.calls["bcast"] += 1
        t = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64).copy())
        dist.broadcast(t, src=root)
        return t.numpy()

    def allgather_records(self, rows, width, capacity_rows):
        import numpy as np, torch.distributed as dist
        self.calls["allgather"] += 1
        box = [None] * self.world_size
        dist.all_gather_object(box, np.asarray(rows, dtype=np.float64).reshape(-1, width))
        out = np.concatenate(box, axis=0)
        assert len(out) <= capacity_rows
        return out


def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch.distributed as dist
    import omc_amd
    bnb = omc_amd.pkg.bnb
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = StubCommEngine(rank, world)
    comm = bnb.Comm(rank, world, eng)
    assert comm.engine is eng                                     # the engine-side communicator is the one used
    # ragged record exchange: rank r contributes r + 1 rows
    rows = np.full((rank + 1, 5), float(rank)) + np.arange(5)[None, :]
    allrows = comm.allgather_rows(rows, 5)
    # incumbent found by the LAST rank (a non-zero owner), broadcast from it
    ub = 10.0 - rank; g_ub, g_lb, owner = comm.min_bounds(ub, float(rank))
    X = np.full((3, 4), float(rank))
    Xg = comm.bcast_matrix(X if rank == owner else np.zeros((3, 4)), owner)
    # a tie: every rank holds the same value -> the smallest rank owns it
    _, _, owner_tie = comm.min_bounds(7.0, 0.0)
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, allrows.tolist(), g_ub, g_lb, owner, Xg.tolist(), owner_tie, eng.calls))


def test_comm_through_engine_communicator_nonzero_owner_and_ragged_records():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    expect_rows = [[float(r) + c for c in range(5)] for r in range(world) for _ in range(r + 1)]
    for rank, allrows, g_ub, g_lb, owner, Xg, owner_tie, calls in res:
        assert allrows == expect_rows                              # rows of rank 0, then rank 1, ... on every rank
        assert g_ub == 10.0 - (world - 1) and g_lb == 0.0 and owner == world - 1
        assert Xg == [[float(world - 1)] * 4] * 3                  # the owner's X arrived everywhere
        assert owner_tie == 0
        assert calls == dict(allreduce=2, bcast=1, allgather=1)
