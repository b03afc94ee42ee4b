"""Randomised parity sweep at the sizes where the tracked-subspace cone kernel is active (n >= 48): HIP vs oracle on random instances,
ranks, cut types and paths.  The oracle solves run in a process pool on the host cores of the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def oracle_job(job):
    os.environ["OMP_NUM_THREADS"] = "1"
    import omc_oracle as orc
    A, mask, k, ct, cuts, rs, q1 = job
    inst = orc.Instance(A, mask, 80.0, k)
    r = orc.sdp_relaxation(inst, cuts, ct, params=orc.RelaxParams(rho_scale=rs, reference_quirk_q1=q1), want_certificate=False)
    return r["termination_status"], r["objective"], r["dual_bound"], r["iters"]


def _indexed(ij):
    return ij[0], oracle_job(ij[1])


def main():
    import multiprocessing as mp
    import omc_amd, omc_oracle as orc
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    rng = np.random.default_rng(seed)
    t0 = time.time(); jobs = []; gpu = []
    ncase = 0
    while time.time() - t0 < budget * 0.25 and ncase < 14:
        n = int(rng.integers(48, 73)); m = n + int(rng.integers(0, 12)); k = int(rng.choice([1, 1, 2]))
        ct = str(rng.choice(["linear", "linear2", "linear3"])); kind = str(rng.choice(["lowrank", "lowrank", "readme"]))
        q1 = bool(rng.integers(0, 2)); rs = float(rng.choice([2.0, 4.0, 8.0]))
        A, mask = orc.make_instance(n, m, k, seed=int(rng.integers(1 << 30)), kind=kind, n_indices=None if kind == "readme" else int(rng.uniform(0.2, 0.5) * n * m), noise=float(rng.choice([0.01, 0.1])))
        eng = omc_amd.Engine(A, mask, 80.0, k)
        P = omc_amd.default_params(rho_scale=rs, reference_quirk_q1=int(q1), breakpoints=int(rng.integers(1, 3)))
        dirs = orc.child_directions(ct, k)
        cuts = []; nodes = [[]]
        for d in range(int(rng.integers(1, 4))):          # the path is grown with the GPU's own results
            o = eng.matrix_completion_SDP_relaxation([cuts], ct, params=P, want_X=False)[0]
            if not o["feasible"]:
                break
            cuts = cuts + [(o["breakpoint_vec"], o["U"], list(dirs[int(rng.integers(len(dirs)))]))]
            nodes.append(list(cuts))
        out = eng.matrix_completion_SDP_relaxation(nodes, ct, params=P, want_X=True)
        st = eng.subspace_stats()
        for c, o in zip(nodes, out):
            jobs.append((A, mask, k, ct, c, rs, q1)); gpu.append((n, m, k, ct, kind, len(c), o, st["calls"]))
        eng.close(); ncase += 1
    print("cases %d nodes %d built in %.0fs; oracle pool..." % (ncase, len(jobs), time.time() - t0), flush=True)
    ref = [None] * len(jobs)
    with mp.get_context("spawn").Pool(14) as pool:
        for q, (i, res) in enumerate(pool.imap_unordered(_indexed, list(enumerate(jobs)), chunksize=1)):
            ref[i] = res
            if q % 8 == 0:
                print("  oracle %d / %d  (%.0fs)" % (q + 1, len(jobs), time.time() - t0), flush=True)      # the GPU box kills a silent command
    bad = 0; worst = 0.0; ndiff_it = 0; nsub = 0
    for (n, m, k, ct, kind, L, o, calls), (rst, robj, rlb, rit) in zip(gpu, ref):
        fin = all(np.isfinite(o[key]).all() for key in ("U", "breakpoint_vec", "lambda_min")) and np.isfinite(o["dual_bound"])
        rel = abs(o["objective"] - robj) / max(1.0, abs(robj)) if o["status_code"] != 3 and rst != 3 else 0.0
        ok = fin and o["status_code"] == rst and (rel <= 2e-6 or rst != 0)
        nsub += calls > 0
        if o["iters"] != rit: ndiff_it += 1
        if rst == 0: worst = max(worst, rel)
        if not ok:
            bad += 1
            print("MISMATCH n %d m %d k %d %s %s L %d: gpu (%s, %.10f, lb %.10f, %d its) oracle (%s, %.10f, lb %.10f, %d its) finite %s" % (n, m, k, ct, kind, L, o["status_code"], o["objective"], o["dual_bound"], o["iters"], rst, robj, rlb, rit, fin), flush=True)
    print("nodes %d (instances with the tracked block active: %d), mismatches %d, worst relative objective difference among certified nodes %.1e, iteration counts differing %d; %.0fs" % (len(jobs), nsub, bad, worst, ndiff_it, time.time() - t0))


if __name__ == "__main__":
    main()
