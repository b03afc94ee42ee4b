"""bench.py -- node-relaxations/sec of the HIP hot path on BASELINE config 2 (100x100, k=1, gamma=80, 20% observed,
linear cuts, smallest_1_eigvec).

A "step" = one pass of the hot path over one batch: B independent B&B nodes (a breadth-first frontier of the config-2 tree, staged
in HBM before the timed region) are relaxed to a certified gap by omc_relax_solve through S <= B slots (continuous batching: a slot
that finishes is harvested and handed the next pending node, so the refill path is part of what is timed).
value = B * steps * n_gpus / max-over-ranks wall time.

Multi-GPU: one process per GPU.  `--gpus N` without a launcher spawns the N ranks itself (before anything touches the GPU); under
torchrun the ranks come from the environment and must match --gpus.  Every rank relaxes ITS OWN shard of a larger frontier (the
subtrees below its round-robin share of the depth-d frontier), so per-GPU work is fixed as N grows (weak scaling); after every step
the ranks exchange min{incumbent UB, open LB} with the library's RCCL all-reduce (C ABI: omc_allreduce_bounds) -- the one exchange
of node-parallel B&B; there is no data-path collective.
"""
import argparse
import faulthandler
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
faulthandler.enable()          # a crash in native code leaves a Python traceback on stderr

F64_PEAK_TFLOPS = 78.6   # AMD MI355X datasheet (vector = matrix fp64); the local guide lists no fp64 peak; measured value: profiles/r02_fp64_peak.json


def f_proj(N):
    """algorithmic flops of one spectral projection of an order-N symmetric matrix."""
    return 13.0 / 3.0 * N ** 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--depth", type=int, default=int(os.environ.get("OMC_BENCH_DEPTH", 11)), help="frontier depth: 2^depth nodes per GPU per step")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("OMC_BENCH_SLOTS", 1024)), help="nodes relaxed concurrently per GPU (continuous batching)")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--accel", type=int, default=int(os.environ.get("OMC_BENCH_ACCEL", 0)), help="1: Anderson acceleration of the ADMM map (library default 0)")
    ap.add_argument("--cpu-nodes", type=int, default=4, help="nodes relaxed by the CPU oracle per leg of cpu_baseline (rank 0, N=1 only)")
    ap.add_argument("--frontier-file", default=None, help="N=1: cache of the tuned penalty scale and the frontier (written when absent, read when present) so that a "
                    "profiled run launches the kernels of the timed steps only (tools/prof_round2.sh)")
    ap.add_argument("--warm", type=int, default=int(os.environ.get("OMC_BENCH_WARM", 1)), help="1: every node starts from its parent's final state (omc_relax_set_warm), as in a B&B run where "
                    "a child is relaxed after its parent -- the ancestors are relaxed level by level before the timed region, each from its own parent's state; "
                    "2: only the frontier's parents are relaxed beforehand, cold; 0: cold starts (the reference rebuilds every model, OMC.jl:1482)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("OMC_BENCH_PIPELINE", 2)), help="1: the K timed steps are K batches handed to the engine as one stream (continuous batching: "
                    "the slots a batch frees while its last nodes converge are refilled from the next batch, as a B&B queue that always holds open nodes would); "
                    "2: the same, but batches 2..K are appended to the RUNNING solve of batch 1 (omc_relax_reserve / omc_relax_append), descriptors uploaded inside the timed region; "
                    "0: each step is staged, solved and drained on its own (rounds 1-2)")
    ap.add_argument("--early-stop", type=float, default=None, help="early_stop_factor of the relaxation parameters (library default 1.5; 0 = rule off)")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-arg", default="{}", help=argparse.SUPPRESS)
    ap.add_argument("--extras", type=int, default=1, help="0: skip latency_b1 / branching / time_to_gap / cpu_baseline (rank 0, N=1 only)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as children (nothing in this process has touched the GPU,
    and the children are fresh interpreters -- never an exec of a process that initialised HIP) and exit with their code."""
    import socket
    import torch
    have = torch.cuda.device_count()          # counts devices without initialising the GPU in this (parent) process
    if have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:                                 # a rank that dies must not leave the others waiting in a collective
        time.sleep(0.5)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = rc or r
                for q in live:
                    q.terminate()
    sys.exit(rc)


def cpu_baseline_legs(A, mask, gamma, k, cut_type, rho_scale, nodes, depth):
    """The CPU restatement of the node relaxation on the host cores of this box, timed AFTER the GPU region (rank 0, N = 1):
    (a) `value`: the compiled single-thread restatement (oracle/omc_cpu_ref.cpp, checked against the numpy oracle in tests/test_cpu_ref.py)
        on one thread -- the way the reference pins its solver (MSK_IPAR_NUM_THREADS = 1, OMC.jl:1486);
    (b) `all_cores`: the same library with OpenMP over independent nodes, one node per host core, every core of the box;
    (c) `numpy_port`: the numpy / LAPACK oracle (oracle/omc_oracle.py) pinned to one thread, as in rounds 1-2.
    Cold starts (the reference rebuilds every model, OMC.jl:1482).  The reference itself (Julia + Mosek) cannot run here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import omc_oracle as orc
    import omc_cpu_ref as cref
    nproc = os.cpu_count() or 1
    inst = orc.Instance(A, mask, gamma, k)
    P = orc.RelaxParams(); P.rho_scale = rho_scale
    t1 = time.perf_counter()
    one, _ = cref.relax_nodes(inst, nodes, cut_type, params=P, threads=1)
    t_one = time.perf_counter() - t1
    share = _cpu_share(nproc)
    workers = max(1, min(nproc, int(os.environ.get("OMC_BENCH_CPU_WORKERS", share))))
    sample = [nodes[i % len(nodes)] for i in range(2 * workers)]
    t1 = time.perf_counter()
    cref.relax_nodes(inst, sample, cut_type, params=P, threads=workers)
    t_all = time.perf_counter() - t1
    t1 = time.perf_counter()
    its_np = _cpu_worker((A, mask, gamma, k, cut_type, rho_scale, nodes[:2]))
    t_np = time.perf_counter() - t1
    return dict(value=len(nodes) / t_one, unit="node-relaxations/s", cores=1, kind="port", nproc=nproc,
                sample=f"first {len(nodes)} nodes of the depth-{depth} frontier, cold starts, compiled single-thread restatement (oracle/omc_cpu_ref.cpp, g++ -O3); "
                       f"iterations {[o['iters'] for o in one]}, status {[o['status_code'] for o in one]}; the reference itself (Julia + Mosek) cannot run here",
                all_cores=dict(value=len(sample) / t_all, cores=workers, sample=f"{len(sample)} nodes (the same {len(nodes)}, repeated), OpenMP over the nodes: one single-threaded node per core, {workers} threads = every core this process may use "
                                      f"(os.cpu_count() = {nproc}; scheduler affinity / cgroup CPU quota = {share}: 256 threads on this box's 16-core share measured 6.6 nodes/s, 28x oversubscribed)"),
                numpy_port=dict(value=len(nodes[:2]) / t_np, cores=1, sample=f"first {len(nodes[:2])} nodes, numpy / LAPACK oracle pinned to one thread (threadpoolctl); iterations {its_np}"))


def branching_extras(gamma, local):
    """The three runs on the instance whose tree really branches (round-based driver, queue-driven driver, Shor valid inequalities): run in a child
    process of bench.py (--child branching), so that nothing in these minute-long driver runs can cost the headline line."""
    import omc_amd
    bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data
    extras = {}
    # ---- an instance whose tree really branches (config 2's own tree closes at the root) ------------------------------------------
    try:
        Ab, maskb = data.branching_instance(seed=0)
        eb = omc_amd.Engine(Ab, maskb, gamma, 1, device=local)
        t1 = time.perf_counter()
        solb, instb = bnb.branch_and_bound(eb, Ab, maskb, gap=1e-4, time_limit=30.0, batch=256, disjunctive_cuts_type="linear")
        tb = time.perf_counter() - t1
        rd = instb["run_details"]
        extras["branching"] = dict(instance="100x100 rank-1 + 0.3 noise, 10% observed (data.branching_instance, seed 0)", sha256=data.instance_sha256(Ab, maskb)[:16],
                                   seconds=tb, gap=solb["gap"], lower_bound=solb["lower_bound"], upper_bound=solb["objective"], nodes_relaxed=rd["nodes_relax_feasible"],
                                   nodes_per_s=rd["nodes_relax_feasible"] / max(rd["solve_time_relaxation"], 1e-9), relaxation_seconds=rd["solve_time_relaxation"],
                                   altmin_seconds=rd["solve_time_altmin"], batch=256, reached_gap=bool(solb["gap"] <= 1e-4))
        # the queue-driven driver on the same instance (one running solve fed by omc_relax_append / omc_relax_fetch_done / omc_relax_hold), same budget
        t1 = time.perf_counter()
        solq, instq = omc_amd.pkg.bnb_stream.branch_and_bound_streaming(eb, Ab, maskb, gap=1e-4, time_limit=30.0, slots=1024, disjunctive_cuts_type="linear")
        rq = instq["run_details"]
        extras["branching_streaming"] = dict(instance="the same instance, bnb_stream.branch_and_bound_streaming (nodes appended to one running solve as their parents finish)",
                                             seconds=time.perf_counter() - t1, gap=solq["gap"], lower_bound=solq["lower_bound"], upper_bound=solq["objective"],
                                             nodes_relaxed=rq["nodes_relax_feasible"], nodes_per_s=rq["nodes_relax_feasible"] / max(rq["solve_time_relaxation"], 1e-9),
                                             solves_staged=rq["epochs"], warm_started=rq["warm_started"], reached_gap=bool(solq["gap"] <= 1e-4))
        # the same instance with the reference's Shor valid inequalities (add_Shor_valid_inequalities = true, static list of the minors with all
        # four entries observed, OMC.jl:646-669): the Shor-mode relaxation closes it at the root
        t1 = time.perf_counter()
        sols, insts = bnb.branch_and_bound(eb, Ab, maskb, gap=1e-4, time_limit=60.0, batch=32, disjunctive_cuts_type="linear", add_Shor_valid_inequalities=True,
                                           Shor_valid_inequalities_noisy_rank1_num_entries_present=(4,),
                                           shor_params=omc_amd.default_params(rho_scale=1.0, eps_gap=1e-5, max_iters=6000, time_limit=30.0))
        ts = time.perf_counter() - t1
        extras["branching_shor"] = dict(instance="the same instance, add_Shor_valid_inequalities = true, minors with four observed entries (static list)",
                                        seconds_to_gap=ts, gap=sols["gap"], lower_bound=sols["lower_bound"], upper_bound=sols["objective"],
                                        nodes_relaxed=insts["run_details"]["nodes_relax_feasible"], reached_gap=bool(sols["gap"] <= 1e-4))
        eb.close()
    except Exception as ex:            # an extra must never cost the headline line
        extras.setdefault("branching", dict(error=repr(ex))); extras.setdefault("branching_streaming", dict(error=repr(ex))); extras.setdefault("branching_shor", dict(error=repr(ex)))
    return extras


def _cpu_share(nproc):
    """Cores this process may really use: the scheduler affinity mask and the cgroup CPU quota (a GPU box of this pool exposes 256 logical CPUs
    and grants a share of them per GPU)."""
    share = nproc
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    share = min(share, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    share = min(share, max(1, q // per))
        except Exception:
            pass
    return max(1, share)


def _cpu_worker(job):
    A, mask, gamma, k, cut_type, rho_scale, nodes = job
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import omc_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    inst = orc.Instance(A, mask, gamma, k)
    its = []
    def run():
        for c in nodes:
            its.append(orc.sdp_relaxation(inst, c, cut_type, params=orc.RelaxParams(rho_scale=rho_scale), want_certificate=False)["iters"])
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            run()
    else:
        run()
    return its


def main():
    args = parse()
    os.environ.setdefault("OMC_SEGV_TRACE", "1")      # a fatal signal inside the library prints its native frames to stderr before the process dies
    if args.child == "branching":
        sys.path.insert(0, ROOT)
        ca = json.loads(args.child_arg)
        print(json.dumps(branching_extras(float(ca.get("gamma", 80.0)), int(ca.get("local", 0)))), flush=True)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)
    world = int(env_world or "1"); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")

    import numpy as np
    import torch
    import torch.distributed as dist
    if world > 1:
        # torch.distributed only carries the rendezvous (the 128-byte RCCL id), the barriers and the MAX of the step times: a gloo group on
        # the host, so that the ONE RCCL instance of the process is the library's own communicator (C ABI), which carries the data path
        torch.cuda.set_device(local)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import omc_amd
    bnb, data = omc_amd.pkg.bnb, omc_amd.pkg.data

    A, mask, gamma, cfg = data.config_instance(args.config, seed=0)
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    eng = omc_amd.Engine(A, mask, gamma, k, device=local)
    use_comm = world > 1 or bool(os.environ.get("OMC_BENCH_FORCE_COMM"))     # the env var exercises the exchange path with a world of one
    lib_comm = False; comm_kind = None
    if use_comm:    # the library's own RCCL communicator (C ABI); torch.distributed only carries the 128-byte id to the ranks
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            torch.cuda.set_device(local)
            dist.init_process_group("gloo", rank=0, world_size=1)
        box = [eng.comm_unique_id().tobytes() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm_kind = "omc_allreduce_bounds (library RCCL communicator)"
        try:
            eng.comm_init(rank, world, np.frombuffer(box[0], dtype=np.uint8))
            ok = 1
        except Exception as e:          # RCCL not loadable / init refused: the 16 bytes then travel over the gloo group (reported in config.bounds_exchange)
            print(f"bench.py rank {rank}: omc_comm_init failed ({e}); falling back to torch.distributed (gloo) all_reduce", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes the same path
        lib_comm = bool(int(flag[0]))
        if not lib_comm:
            comm_kind = "torch.distributed all_reduce (gloo, host); omc_comm_init failed"
    cache = args.frontier_file if world == 1 else None
    if cache and os.path.exists(cache):
        import pickle
        with open(cache, "rb") as f:                 # a file this script wrote itself
            saved = pickle.load(f)
        if saved["key"] != (args.config, args.depth):
            sys.exit(f"bench.py: {cache} holds config/depth {saved['key']}")
        rho_scale, tune_log, nodes = saved["rho_scale"], saved["tune_log"], saved["nodes"]
        P = omc_amd.default_params(rho_scale=rho_scale, slots=args.slots, accel=args.accel, **({} if args.early_stop is None else {"early_stop_factor": args.early_stop}))
    else:
        rho_scale, tune_log = bnb.autotune_rho_scale(eng, cfg["cut_type"])
        P = omc_amd.default_params(rho_scale=rho_scale, slots=args.slots, accel=args.accel, **({} if args.early_stop is None else {"early_stop_factor": args.early_stop}))
        # every rank builds the same depth-d frontier (deterministic), keeps its round-robin share and expands THOSE subtrees until it
        # holds 2^d nodes again: distinct work per rank, identical amount of it
        nodes, _ = bnb.expand_frontier(eng, args.depth, cfg["cut_type"], params=P)
        if cache:
            import pickle
            with open(cache, "wb") as f:
                pickle.dump(dict(key=(args.config, args.depth), rho_scale=rho_scale, tune_log=tune_log, nodes=nodes), f)
    B = len(nodes)
    if world > 1:
        mine = bnb.shard_nodes(nodes, rank, world)
        while len(mine) < B:
            out = eng.matrix_completion_SDP_relaxation(mine, cfg["cut_type"], params=P, want_Y=False, want_X=False)
            nxt = []
            for cuts, o in zip(mine, out):
                nxt.extend(bnb.make_children(cuts, o, cfg["cut_type"], k) if o["feasible"] else [cuts])
            mine = nxt
        nodes = mine[:B]
    # warm start: a node of the frontier is a child of a depth-(d-1) node, itself the child of a depth-(d-2) node, ...  As in a B&B run, where
    # every node inherits its parent's final state (omc_relax_set_warm), the ancestors are relaxed level by level (untimed), each from its
    # own parent's state, and leave their final states in the device pool -- the depth-(d-1) states are inputs of the timed region, as the
    # parent's state is an input of a child's relaxation in the driver.  --warm 2: the depth-(d-1) parents alone, cold-started (round-3 first form)
    load_from = None; n_parents = 0
    if args.warm:
        pidx = {(): 0}; levels = [[((), [])]]                  # prefix key (ids of the cut objects) -> pool index ; per depth: (key, cut list)
        for cuts in nodes:
            for l in range(1, len(cuts)):
                key = tuple(id(c) for c in cuts[:l])
                if key not in pidx:
                    pidx[key] = len(pidx)
                    while len(levels) <= l:
                        levels.append([])
                    levels[l].append((key, list(cuts[:l])))
        load_from = [pidx[tuple(id(c) for c in cuts[:-1])] if len(cuts) else -1 for cuts in nodes]
        n_parents = len(levels[-1])
        eng.state_pool_create(len(pidx))
        for l, lev in enumerate(levels):
            if args.warm == 2 and l + 1 < len(levels):
                continue
            lf = [-1 if (l == 0 or args.warm == 2) else pidx[key[:-1]] for key, _ in lev]
            eng.matrix_completion_SDP_relaxation([c for _, c in lev], cfg["cut_type"], params=P, want_Y=False, want_X=False, load_from=lf, save_to=[pidx[key] for key, _ in lev])
    eng.stage(nodes, cfg["cut_type"], P, load_from=load_from)           # node descriptors (and parent states) resident in HBM before the timed region

    def exchange(out):
        ub = min(o["objective"] for o in out); lb = min(o["dual_bound"] for o in out)
        if use_comm and lib_comm:
            ub, lb, _ = eng.allreduce_bounds(ub, lb)
        elif use_comm:
            t = torch.tensor([ub, lb], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ub, lb = float(t[0]), float(t[1])
        return ub, lb

    def step():
        eng.solve()
        out = eng.fetch(want_Y=False, want_X=False)
        return exchange(out), out

    # every launch of the timed steps is bracketed by HIP events: hipGraph replay of the iteration body (the library's path for <= 16 live
    # slots, whose kernels events cannot see) is switched off for them, so that the per-kernel averages below and rocprofv3's describe the
    # same launches; the extras (latency_b1, branching) run with the library default
    if os.environ.get("OMC_TIMING_STRIDE", "1") == "1":
        eng.tuning_set("OMC_GRAPH_MAX", "0")
    for _ in range(args.warmup):
        step()
    kstats = {}; sub = {}
    markers = bool(os.environ.get("OMC_BENCH_MARKERS"))          # profiling runs: one k_eval_objective launch before and one after the timed steps
    if markers:                                                   # (nothing else in this command launches it), so that a kernel trace can be cut to them
        eng.evaluate_objective(A)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pipelined = bool(args.pipeline) and args.steps > 1
    appended = pipelined and args.pipeline == 2
    if appended:       # --pipeline 2: the first batch is staged with room for the others, which are APPENDED to the running solve (omc_relax_append)
        eng.reserve(B * (args.steps - 1), max(len(c) for c in nodes))
        eng.stage(nodes, cfg["cut_type"], P, load_from=load_from)
        torch.cuda.synchronize()
    elif pipelined:    # the K batches as one stream: node b of batch s is node s * B + b (descriptors and parent states resident before the clock starts)
        eng.stage(nodes * args.steps, cfg["cut_type"], P, load_from=None if load_from is None else load_from * args.steps)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(1 if pipelined else args.steps):
        if pipelined:
            if appended:
                eng.submit()
                for q in range(1, args.steps):                    # host -> device of the descriptors is inside the timed region here
                    eng.append(nodes, cfg["cut_type"], load_from=load_from)
                eng.wait()
            else:
                eng.solve()
            out_all = eng.fetch(want_Y=False, want_X=False)
            for q in range(args.steps):                      # one bound exchange per batch, as in the step-by-step form
                exchange(out_all[q * B:(q + 1) * B])
            out = out_all[(args.steps - 1) * B:]
        else:
            _, out = step()
        for c, v in eng.kernel_stats().items():
            a = kstats.setdefault(c, dict(launches=0, ms=0.0, units=0))
            a["launches"] += v["launches"]; a["ms"] += v["ms"]; a["units"] += v["units"]
        for c, v in eng.subspace_stats().items():
            sub[c] = sub.get(c, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if markers:
        eng.evaluate_objective(A)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])
    info = eng.solver_info()
    value_all = B * args.steps * world / el
    iters = np.array([o["iters"] for o in out]); status = np.bincount([o["status_code"] for o in out], minlength=4)
    certified = int(status[0] + status[3])
    # headline = nodes returned with a certificate (two-sided gap 1e-6, or proven infeasible) per second; SLOW_PROGRESS nodes come back
    # with values and a valid bound (MOI.SLOW_PROGRESS, OMC.jl:1871-1877) and are counted in value_all only
    value = value_all * certified / B

    # ---- roofline of the cone block (the kernel class VERDICT r01 named): batched order-n spectral projection = k_cone_sub (tracked
    # subspace, MFMA) + k_cone_ws (full eigendecomposition: seed / fall-back).  Algorithmic flops per projection F_proj(n)
    # x projections / the HIP-event time of both kernels (events on the stream they are launched on).
    cone_ms = kstats["cone"]["ms"] + kstats["cone_sub"]["ms"]
    projections = kstats["global"]["units"]                      # one projection per node-iteration
    launches = max(1, kstats["global"]["launches"])
    achieved = f_proj(n) * projections / (max(cone_ms, 1e-9) * 1e-3) / 1e12          # cone_ms = 0 only with OMC_TIMING_STRIDE=0 (no event timing: experiments)
    np16 = (n + 15) // 16 * 16
    executed_sub = (sub.get("power_steps", 0) + sub.get("ritz_passes", 0)) * 2.0 * np16 * np16 * 16     # MFMA flops of the M X products of k_cone_sub
    traffic = None; peak_measured = None
    try:
        with open(os.path.join(ROOT, "profiles", "r03_cone_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_cone_traffic.json")) else "r02_cone_traffic.json")) as f:
            traffic = json.load(f)             # PMC passes are separate rocprofv3 runs (tools/prof_round3.sh); the file says which regime it describes
        with open(os.path.join(ROOT, "profiles", "r02_fp64_peak.json")) as f:
            peak_measured = json.load(f)
    except Exception:
        pass
    per_kernel = {c: dict(ms=round(v["ms"], 2), launches=v["launches"], avg_launch_ms=round(v["ms"] / max(1, v["launches"]), 4)) for c, v in kstats.items()}
    # one roofline per iteration kernel: algorithmic work per node-iteration x node-iterations / the kernel's HIP-event time
    cj = mask.sum(0).astype(float)
    f_frac = float((cj ** 3 / 3.0 + 2.0 * cj ** 2).sum())                     # K-FRAC: one factorisation + solves per observed column
    b_frac = 8.0 * n * m + n * m / 8.0 + 16.0 * n * n                         # A, mask, Y in, gradient out
    b_glob = 96.0 * n * n                                                     # k_global: Y, D1, D3, W1, E3, weights in; Y, Yp, D1, D3, next cone input out (12 n^2 doubles)
    r_eff = max(1, int(info["r_max"]))
    f_small = 4.0 * n * n * r_eff + 13.0 / 3.0 * (r_eff + k) ** 3
    HBM_PEAK = 8.0                                                            # TB/s, MI355X_MICROARCH.md
    def kroof(name, cls, bound, work, peak, unit):
        ms_ = sum(kstats[c]["ms"] for c in cls)
        ach = work * projections / max(ms_ * 1e-3, 1e-12) / 1e12
        return dict(kernel=name, bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, ms=round(ms_, 2), work_per_node_iteration=work)
    kernel_rooflines = [
        kroof("k_colprox_pair (two columns per wave; k_colprox for columns above 32 rows)", ["colprox"], "mfma", f_frac, F64_PEAK_TFLOPS, "TFLOP/s"),
        kroof("k_colprox_pair (bytes)", ["colprox"], "hbm", b_frac, HBM_PEAK, "TB/s"),
        kroof("cone block: k_cone_sub + k_cone_ws", ["cone", "cone_sub"], "mfma", f_proj(n), F64_PEAK_TFLOPS, "TFLOP/s"),
        kroof("k_global", ["global"], "hbm", b_glob, HBM_PEAK, "TB/s"),
        kroof("k_small", ["small"], "mfma", f_small, F64_PEAK_TFLOPS, "TFLOP/s"),
    ]
    roofline = dict(bound="mfma", kernel="cone block: k_cone_sub + k_cone_ws", achieved=achieved, peak=F64_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / F64_PEAK_TFLOPS,
                    traffic=traffic, peak_measured=peak_measured, avg_launch_ms=cone_ms / launches, matrices_per_launch=projections / launches, order=n,
                    note="algorithmic flops F_proj(n) = 13/3 n^3 per projection (SURVEY 8d) over the summed HIP-event time of both cone kernels; "
                         "k_cone_sub replaces the eigendecomposition by MFMA power steps on a 16-vector block, so the flops it EXECUTES are fewer "
                         "(executed_subspace_tflops) -- the algorithmic figure measures the time per projection, not MFMA utilisation",
                    executed_subspace_tflops=executed_sub / max(kstats["cone_sub"]["ms"] * 1e-3, 1e-12) / 1e12, subspace=sub,
                    concurrency="the cone kernels share the CUs with k_colprox and k_small of the same iteration (three HIP streams); OMC_STREAMS=1 runs "
                                "the kernels back to back (profiles/ has both)",
                    kernel_ms=per_kernel, kernels=kernel_rooflines)

    eng.tuning_reload_env()                        # the extras run with the library default (or whatever the caller's environment says)
    extras = {}
    if rank == 0 and world == 1 and args.extras:
        # ---- single-node-at-a-time (BASELINE config 2 wording): batch 1, the reference's serial order --------------------------------
        sample = nodes[:: max(1, B // 12)][:12]
        e1 = omc_amd.Engine(A, mask, gamma, k, device=local)
        P1 = omc_amd.default_params(rho_scale=rho_scale, slots=1, accel=args.accel)
        e1.matrix_completion_SDP_relaxation([sample[0]], cfg["cut_type"], params=P1, want_Y=False, want_X=False)
        t1 = time.perf_counter(); its1 = []
        for c in sample:
            its1.append(e1.matrix_completion_SDP_relaxation([c], cfg["cut_type"], params=P1, want_Y=False, want_X=False)[0]["iters"])
        tl = time.perf_counter() - t1
        e1.close()
        extras["latency_b1"] = dict(value=len(sample) / tl, unit="node-relaxations/s", nodes=len(sample), ms_per_node=tl / len(sample) * 1e3,
                                    iters_mean=float(np.mean(its1)), us_per_iteration=tl / max(1, sum(its1)) * 1e6)
        # ---- an instance whose tree really branches (config 2's own tree closes at the root): three driver runs, in a child process ------------
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "branching", "--child-arg", json.dumps(dict(gamma=gamma, local=local))],
                                capture_output=True, text=True, timeout=400)
            line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            if cp.returncode == 0 and line:
                extras.update(json.loads(line[-1]))
            else:
                err = dict(error=f"child exited with {cp.returncode}", stderr_tail=cp.stderr[-1500:])
                for key in ("branching", "branching_streaming", "branching_shor"):
                    extras[key] = err
        except Exception as ex:            # an extra must never cost the headline line
            for key in ("branching", "branching_streaming", "branching_shor"):
                extras[key] = dict(error=repr(ex))
        # ---- second half of the metric: wall-clock of the whole B&B (root altmin, penalty autotune, tree) to gap <= 1e-4 ---------
        tt = []
        for sd in (0, 1, 2):
            A2, mask2, g2, c2 = data.config_instance(args.config, seed=sd)
            e2 = omc_amd.Engine(A2, mask2, g2, c2["k"], device=local)
            t1 = time.perf_counter()
            sol, inst2 = bnb.branch_and_bound(e2, A2, mask2, gap=1e-4, time_limit=120.0, batch=128, disjunctive_cuts_type=c2["cut_type"])
            tt.append(dict(seed=sd, seconds=time.perf_counter() - t1, gap=sol["gap"], nodes_relaxed=inst2["run_details"]["nodes_relax_feasible"]))
            e2.close()
        extras["time_to_gap"] = dict(target_gap=1e-4, runs=tt, median_seconds=float(np.median([t_["seconds"] for t_ in tt])))
        # ---- the same on BASELINE config 1 (README quick-start: 50 x 50 pure noise, gamma = 80, bestfirst + linear cuts): a tree that branches ----
        try:
            A1, mask1, g1, c1 = data.config_instance(1, seed=0)
            e1 = omc_amd.Engine(A1, mask1, g1, c1["k"], device=local)
            t1 = time.perf_counter()
            sol1, inst1 = bnb.branch_and_bound(e1, A1, mask1, gap=1e-4, time_limit=float(os.environ.get("OMC_BENCH_CFG1_SECONDS", 45)), batch=256, disjunctive_cuts_type=c1["cut_type"])
            tb1 = time.perf_counter() - t1
            rd1 = inst1["run_details"]; lg = inst1["run_log"]
            traj = [dict(seconds=round(r_[6], 2), explored=int(r_[0]), lower=r_[3], upper=r_[4], gap=r_[5]) for r_ in lg[:: max(1, len(lg) // 8)]]
            extras["time_to_gap_config1"] = dict(instance="config 1: README quick-start 50x50, A iid N(0,1), mask Bernoulli(1/2), seed 0", seconds=tb1, reached_gap=bool(sol1["gap"] <= 1e-4),
                                                  gap=sol1["gap"], lower_bound=sol1["lower_bound"], upper_bound=sol1["objective"], nodes_relaxed=rd1["nodes_relax_feasible"],
                                                  nodes_per_s=rd1["nodes_relax_feasible"] / max(rd1["solve_time_relaxation"], 1e-9), trajectory=traj)
            e1.close()
        except Exception as ex:
            extras["time_to_gap_config1"] = dict(error=repr(ex))
        # ---- BASELINE config 3 as defined: 200 x 200 rank 1 with add_Shor_valid_inequalities = true, static class-4 list (Shor mode) ------------
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            A3, mask3, g3, c3 = data.config_instance(3, seed=0)
            e3 = omc_amd.Engine(A3, mask3, g3, 1, device=local)
            mi3 = e3.generate_rank1_matrix_completion_Shor_constraints_indexes([4])
            nb3 = int(os.environ.get("OMC_BENCH_SHOR_NODES", 8)); it3 = int(os.environ.get("OMC_BENCH_SHOR_ITERS", 400))
            e3.tuning_set("OMC_GRAPH_MAX", "0")
            e3.stage_shor([[]] * nb3, [(mi3, None)] * nb3, "linear", omc_amd.default_params(max_iters=it3, slots=nb3, eps_gap=1e-5))
            t1 = time.perf_counter(); e3.solve(); ts3 = time.perf_counter() - t1
            o3 = e3.fetch(want_Y=False, want_X=False)
            ks3 = {c_: round(v_["ms"], 1) for c_, v_ in e3.kernel_stats().items() if v_["launches"]}
            extras["shor_config3"] = dict(workload=f"config 3: 200x200 rank-1, {len(mi3)} class-4 minors (device enumeration), {nb3} root copies, {it3} ADMM iterations each (a root certifies 1e-5 in ~2550: tests/test_gpu_shor.py)",
                                          minors=int(len(mi3)), node_iterations_per_s=nb3 * int(o3[0]["iters"]) / ts3, ms_per_iteration=ts3 / max(1, int(o3[0]["iters"])) * 1e3,
                                          kernel_ms=ks3, big_cone_order=int(A3.shape[0] + A3.shape[1]),
                                          big_cone_tflops=f_proj(A3.shape[0] + A3.shape[1]) * nb3 * int(o3[0]["iters"]) / max(ks3.get("shor_bigcone", 0.0) * 1e-3, 1e-12) / 1e12)
            e3.close()
        except Exception as ex:
            extras["shor_config3"] = dict(error=repr(ex))
        # ---- the same frontier from cold starts (what every round before round 3 measured) -------------------------------------------------
        if args.warm:
            try:
                eng.stage(nodes, cfg["cut_type"], P)
                t1 = time.perf_counter(); eng.solve(); tc = time.perf_counter() - t1
                oc = eng.fetch(want_Y=False, want_X=False)
                stc = np.bincount([o_["status_code"] for o_ in oc], minlength=4)
                extras["cold_start"] = dict(nodes_per_s_all=B / tc, certified_nodes_per_s=float(stc[0] + stc[3]) / tc, certified_fraction=float(stc[0] + stc[3]) / B,
                                            iters_median=int(np.median([o_["iters"] for o_ in oc])), seconds=tc)
            except Exception as ex:
                extras["cold_start"] = dict(error=repr(ex))
            # ---- and the warm frontier as ONE batch drained to its last node (no next batch behind it): the per-step form of rounds 1-2 ------------
            try:
                eng.stage(nodes, cfg["cut_type"], P, load_from=load_from)
                t1 = time.perf_counter(); eng.solve(); td = time.perf_counter() - t1
                od = eng.fetch(want_Y=False, want_X=False)
                std = np.bincount([o_["status_code"] for o_ in od], minlength=4)
                extras["one_batch_drained"] = dict(nodes_per_s_all=B / td, certified_nodes_per_s=float(std[0] + std[3]) / td, seconds=td,
                                                   note="the same 2048 warm-started nodes as one batch run to its last node; the headline streams K such batches (config.pipelined_batches)")
            except Exception as ex:
                extras["one_batch_drained"] = dict(error=repr(ex))
        if args.cpu_nodes > 0:
            extras["cpu_baseline"] = cpu_baseline_legs(A, mask, gamma, k, cfg["cut_type"], rho_scale, nodes[: args.cpu_nodes], args.depth)
    if rank == 0:
        print(json.dumps({
            "metric": "B&B node-relaxations/sec, 100x100 k=1", "value": value, "value_note": "nodes returned with a certificate per second; all nodes (certified + SLOW_PROGRESS with values and a valid bound): config.nodes_per_s_all", "unit": "node-relaxations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config {args.config}: {n}x{m} rank-{k}, gamma=80, 20% observed, {cfg['cut_type']} cuts, smallest_1_eigvec; "
                                   f"{B} depth-{args.depth} frontier nodes per GPU per step streamed through {min(args.slots, B)} slots (continuous batching, "
                                   + ("every node warm-started from its parent's final state" if args.warm else "cold starts")
                                   + (", each rank its own subtrees" if world > 1 else "")
                                   + (f"; the {args.steps} steps are {args.steps} batches handed over as one stream: slots freed while a batch's last nodes converge are refilled from the next batch"
                                      + (" (batches 2.. are appended to the running solve of batch 1: omc_relax_append, descriptor upload inside the timed region)" if appended else "") if pipelined else "") + ")",
                       "nodes_per_gpu": B, "slots": min(args.slots, B), "pipelined_batches": pipelined, "appended_to_running_solve": appended, "warm_start": bool(args.warm), "parent_states_in_pool": n_parents, "rho_scale": rho_scale, "eps_gap": 1e-6, "iters_median": int(np.median(iters)), "iters_max": int(iters.max()),
                       "status_counts": {"optimal": int(status[0]), "slow_progress": int(status[1]), "time_limit": int(status[2]), "infeasible": int(status[3])},
                       "certified_fraction": certified / B, "certified_nodes_per_s": value, "nodes_per_s_all": value_all,
                       "bounds_exchange": (comm_kind if use_comm else None),
                       "jacobi_sweeps_last_step": info["jacobi_sweeps"], "instance_sha256": data.instance_sha256(A, mask)[:16]},
            "roofline": roofline, "cpu_baseline": extras.get("cpu_baseline"), "time_to_gap": extras.get("time_to_gap"),
            "latency_b1": extras.get("latency_b1"), "branching": extras.get("branching"), "branching_shor": extras.get("branching_shor"), "branching_streaming": extras.get("branching_streaming"), "time_to_gap_config1": extras.get("time_to_gap_config1"),
            "shor_config3": extras.get("shor_config3"), "cold_start": extras.get("cold_start"), "one_batch_drained": extras.get("one_batch_drained"),
        }))
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
