"""Randomised parity sweep: small random instances (sizes, ranks, cut types, masks, paths) -- HIP path vs the oracle.
Reports any node whose status differs or whose certified objective differs by more than 2e-6 relative."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, omc_amd
import omc_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
t0 = time.time(); ncmp = nbad = ninst = 0; worst = 0.0
while time.time() - t0 < budget:
    n = int(rng.integers(6, 22)); m = n + int(rng.integers(0, 10)); k = int(rng.choice([1, 1, 2]))
    kind = str(rng.choice(["lowrank", "readme"])); ct = str(rng.choice(["linear", "linear2", "linear3"]))
    frac = float(rng.uniform(0.3, 0.9))
    try:
        A, mask = orc.make_instance(n, m, k, seed=int(rng.integers(1 << 30)), kind=kind, n_indices=max(int(frac * n * m), (n + m) * k))
    except ValueError:
        continue
    rs = float(rng.choice([1.0, 4.0, 16.0]))
    inst = orc.Instance(A, mask, 80.0, k); eng = omc_amd.Engine(A, mask, 80.0, k); ninst += 1
    dirs = orc.child_directions(ct, k)
    cuts = []; nodes = [[]]
    for d in range(int(rng.integers(1, 4))):
        r = orc.sdp_relaxation(inst, cuts, ct, params=orc.RelaxParams(rho_scale=rs, max_iters=1500), want_certificate=False)
        if r["termination_status"] == 3: break
        x, ev = orc.breakpoint_vector(r["Y"], r["U"])
        cuts = cuts + [(x, r["U"].copy(), list(dirs[int(rng.integers(len(dirs)))]))]
        nodes.append(list(cuts))
    out = eng.matrix_completion_SDP_relaxation(nodes, ct, params=omc_amd.default_params(rho_scale=rs, max_iters=1500), want_X=False)
    for cset, g in zip(nodes, out):
        r = orc.sdp_relaxation(inst, cset, ct, params=orc.RelaxParams(rho_scale=rs, max_iters=1500), want_certificate=False)
        ncmp += 1
        ok = g["status_code"] == r["termination_status"]
        if ok and g["status_code"] == 0:
            rel = abs(g["objective"] - r["objective"]) / max(1.0, abs(r["objective"])); worst = max(worst, rel)
            ok = rel <= 2e-6 and abs(g["iters"] - r["iters"]) <= 25
        if not ok:
            nbad += 1
            print("MISMATCH n %d m %d k %d %s %s L %d rs %.0f: gpu (%d, %d its, %.9f) oracle (%d, %d its, %.9f)" % (n, m, k, kind, ct, len(cset), rs, g["status_code"], g["iters"], g["objective"], r["termination_status"], r["iters"], r["objective"]), flush=True)
    eng.close()
print("instances %d, nodes compared %d, mismatches %d, worst relative objective difference among certified nodes %.2e, %.0fs" % (ninst, ncmp, nbad, worst, time.time() - t0))
