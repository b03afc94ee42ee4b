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
